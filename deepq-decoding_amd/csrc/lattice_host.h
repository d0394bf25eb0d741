// Host-side surface-code lattice tables for any odd distance (the closed forms of oracle/lattice.py; reference:
// /root/reference/cluster_scripts/d5_dp/Function_Library.py:13-51 generateSurfaceCodeLattice, :189-221 measurement order,
// :312-317 logical operators; example_notebooks/Environments.py:262-385 neighbourhoods and embeddings).  Used by match.hip and env_big.hip;
// env.hip keeps its own 64-bit-word builder for d <= 7.
#pragma once
#include <vector>
#include "common.h"

struct LatticeHost {
    int d, d2, n_stab, n_side;                     // n_side = 2d + 1
    std::vector<int> sa, sb, stab_type;            // plaquette (a, b) and type (3: X component, 1: Z component) of stabilizer s, measurement order
    std::vector<int> index;                        // [(d+1)*(d+1)] plaquette -> stabilizer, -1: absent
    std::vector<std::vector<int>> stab_qubits;     // qubits of stabilizer s
    std::vector<std::vector<int>> qubit_stabs;     // live stabilizers touched by qubit q (ENV:262-271)
    std::vector<std::vector<int>> neigh;           // 8-neighbourhood of qubit q (ENV:349-372)
    std::vector<int> ref_bit;                      // rank of stabilizer s among the plaquettes of its own type in row-major (a, b) order
    std::vector<int> typed[2];                     // comp 0 (type 3) / comp 1 (type 1): stabilizers in that order
};

static inline int lattice_plaquette_type(int d, int a, int b) {     // FL:32-35, FL:42-50
    if ((a == 0 && b % 2 == 0) || (a == d && b % 2 == 1) || (b == 0 && a % 2 == 1) || (b == d && a % 2 == 0)) return 0;
    return ((a + b) & 1) ? 3 : 1;
}

static inline void lattice_build(int d, LatticeHost* L) {
    L->d = d; L->d2 = d * d; L->n_stab = d * d - 1; L->n_side = 2 * d + 1;
    const int half = (d + 1) / 2 - 1, ns = L->n_stab;
    L->sa.clear(); L->sb.clear();
    for (int a = 1; a < d; ++a) for (int b = 1; b < d; ++b) { L->sa.push_back(a); L->sb.push_back(b); }      // FL:189-194
    for (int x = 0; x < half; ++x) { L->sa.push_back(0); L->sb.push_back(2 * x + 1); }                        // FL:197-202
    for (int x = 0; x < half; ++x) { L->sa.push_back(d); L->sb.push_back(2 * x + 2); }                        // FL:203-208
    for (int x = 0; x < half; ++x) { L->sa.push_back(2 * x + 2); L->sb.push_back(0); }                        // FL:210-215
    for (int x = 0; x < half; ++x) { L->sa.push_back(2 * x + 1); L->sb.push_back(d); }                        // FL:216-221
    L->index.assign((d + 1) * (d + 1), -1);
    L->stab_type.assign(ns, 0);
    L->stab_qubits.assign(ns, {});
    L->qubit_stabs.assign(d * d, {});
    L->neigh.assign(d * d, {});
    for (int s = 0; s < ns; ++s) {
        const int a = L->sa[s], b = L->sb[s];
        L->index[a * (d + 1) + b] = s;
        L->stab_type[s] = lattice_plaquette_type(d, a, b);
        for (int x = a - 1; x <= a; ++x) for (int y = b - 1; y <= b; ++y)
            if (x >= 0 && x < d && y >= 0 && y < d) { L->stab_qubits[s].push_back(x * d + y); L->qubit_stabs[x * d + y].push_back(s); }
    }
    L->ref_bit.assign(ns, 0);
    L->typed[0].clear(); L->typed[1].clear();
    for (int a = 0; a <= d; ++a) for (int b = 0; b <= d; ++b) {
        const int t = lattice_plaquette_type(d, a, b);
        if (t == 0) continue;
        const int s = L->index[a * (d + 1) + b], comp = t == 3 ? 0 : 1;
        L->ref_bit[s] = (int)L->typed[comp].size();
        L->typed[comp].push_back(s);
    }
    for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c)
        for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {
            const int rr = r + dr, cc = c + dc;
            if ((dr || dc) && rr >= 0 && rr < d && cc >= 0 && cc < d) L->neigh[r * d + c].push_back(rr * d + cc);
        }
}

// Matching-referee tables of one component (oracle/matching_referee.py ComponentGraph): dist [n][n][2], distB [n][2], w10.
static inline void lattice_match_tables(const LatticeHost& L, int comp, std::vector<u8>* dist, std::vector<u8>* distB, int* w10) {
    const int d = L.d, n = (int)L.typed[comp].size(), typ = comp == 0 ? 3 : 1;
    struct Edge { int to, lg; };
    std::vector<std::vector<Edge>> adj(n);
    for (int x = 0; x < d; ++x) for (int y = 0; y < d; ++y) {
        int ends[4], ne = 0;
        for (int s : L.qubit_stabs[x * d + y]) if (L.stab_type[s] == typ) ends[ne++] = L.ref_bit[s];
        const int lg = comp == 0 ? (y == 0) : (x == 0);                                                          // FL:312-317
        if (ne == 2) { adj[ends[0]].push_back({ends[1], lg}); adj[ends[1]].push_back({ends[0], lg}); }
        else if (ne == 1) adj[ends[0]].push_back({-1, lg});
    }
    dist->assign((size_t)n * n * 2, 255);
    distB->assign((size_t)n * 2, 255);
    std::vector<int> seen(2 * n), frontier, next;
    for (int u = 0; u < n; ++u) {
        std::fill(seen.begin(), seen.end(), -1);
        seen[2 * u] = 0;
        frontier.assign(1, 2 * u);
        int w = 0;
        while (!frontier.empty()) {
            ++w;
            next.clear();
            for (int xc : frontier) {
                const int x = xc >> 1, c = xc & 1;
                for (const Edge& e : adj[x]) {
                    const int c2 = c ^ e.lg;
                    if (e.to < 0) { if ((*distB)[2 * u + c2] == 255) (*distB)[2 * u + c2] = (u8)w; }             // the boundary ends a path
                    else if (seen[2 * e.to + c2] < 0) { seen[2 * e.to + c2] = w; next.push_back(2 * e.to + c2); }
                }
            }
            frontier.swap(next);
        }
        for (int yc = 0; yc < 2 * n; ++yc) if (seen[yc] >= 0) (*dist)[(size_t)u * n * 2 + yc] = (u8)seen[yc];
    }
    int best = 255;
    for (int u = 0; u < n; ++u) {
        for (const Edge& e : adj[u]) if (e.to < 0) { const int v = 1 + (*distB)[2 * u + (1 ^ e.lg)]; if (v < best) best = v; }
        if ((*dist)[((size_t)u * n + u) * 2 + 1] < best) best = (*dist)[((size_t)u * n + u) * 2 + 1];
    }
    *w10 = best;
}
