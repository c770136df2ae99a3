"""Exhaustive check of fft_engine.h's lds_delta: for every radix plan of the library, every stage, every butterfly j and every
leg i, the padded LDS index of the accessed element equals the padded index of the butterfly's first element plus a compile-time
constant (reads: j + i * N/R; writes: j0 + i * NS).  Prints the number of violations (0)."""
plans = {16: [[16]], 32: [[8, 4]], 64: [[8, 8]], 128: [[16, 8]], 256: [[16, 16]], 512: [[8, 8, 8]], 1024: [[16, 8, 8], [8, 8, 4, 4]],
         2048: [[16, 16, 8], [8, 8, 8, 4]], 4096: [[16, 16, 16], [8, 8, 8, 8]]}
pad = lambda x: x + x // 16
bad = 0
for n, pls in plans.items():
    for plan in pls:
        ns = 1
        for r in plan:
            s = n // r
            for j in range(n // r):
                k = j % ns
                j0 = (j - k) * r + k
                for i in range(r):
                    bad += pad(j0 + i * ns) != pad(j0) + pad(i * ns)
                    bad += pad(j + i * s) != pad(j) + pad(i * s)
            ns *= r
print("violations:", bad)
