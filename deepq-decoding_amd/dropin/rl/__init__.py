"""Drop-in for the keras-rl fork's `rl` package (the names the reference's scripts import)."""
