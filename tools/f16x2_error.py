"""Representation error of the fused chains' operand schemes against float64, on contractions shaped like this network's layers
(development aid, CPU only; csrc/qnet.h "f16x2").

    python tools/f16x2_error.py

f16x2: x = f16(x) + 2^-11 f16((x - f16(x)) 2^11), products hH + 2^-11 (hL + lH);  bf16x6: round 1's exact three-way bf16 split, six
products;  bf16x3: the three leading bf16 products only;  sgemm-f32: numpy's ordinary float32 matrix product of the same operands.
Piece products are exact in f32, so they are accumulated here in float64: what is shown is the scheme's own error, without the f32
accumulation error every f32 implementation adds on top."""
import numpy as np
rng = np.random.default_rng(0)
def split_f16(x, scale=2048.0):
    h = x.astype(np.float16).astype(np.float32)
    l = ((x - h) * np.float32(scale)).astype(np.float16).astype(np.float32)
    return h, l
def split_bf16x3(x):
    def trunc(v):
        return (v.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    h = trunc(x); r = x - h; m = trunc(r); l = trunc(r - m)
    return h, m, l
for K, M, N, wscale, name in [(1152, 512, 512, 0.03, "dense1"), (256, 2048, 32, 0.06, "conv2"), (63, 4096, 64, 0.1, "conv1"), (512, 512, 51, 0.05, "head")]:
    a = np.maximum(rng.standard_normal((M, K)).astype(np.float32) * 0.7, 0)
    if name == "conv1": a = (rng.random((M, K)) < 0.1).astype(np.float32)
    w = (rng.standard_normal((K, N)) * wscale).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    f32 = a @ w       # numpy sgemm (blocked f32 accumulate)
    # naive sequential f32 accumulation
    ah, al = split_f16(a); wh, wl = split_f16(w)
    d = lambda x: x.astype(np.float64)
    f16x2 = d(ah) @ d(wh) + (d(ah) @ d(wl) + d(al) @ d(wh)) / 2048.0
    bh, bm, bl = split_bf16x3(a); vh, vm, vl = split_bf16x3(w)
    bf6 = d(bh)@d(vh) + d(bh)@d(vm) + d(bm)@d(vh) + d(bm)@d(vm) + d(bh)@d(vl) + d(bl)@d(vh)
    bf3 = d(bh)@d(vh) + d(bh)@d(vm) + d(bm)@d(vh)
    sc = np.abs(ref).max()
    print(f"{name:7s} K={K:5d} max|ref|={sc:.3f}  max abs err: sgemm-f32 {np.abs(f32-ref).max():.2e}  f16x2(exact acc) {np.abs(f16x2-ref).max():.2e}  bf16x6 {np.abs(bf6-ref).max():.2e}  bf16x3 {np.abs(bf3-ref).max():.2e}   rms: f32 {np.sqrt(((f32-ref)**2).mean()):.2e} f16x2 {np.sqrt(((f16x2-ref)**2).mean()):.2e} bf6 {np.sqrt(((bf6-ref)**2).mean()):.2e}")
