"""Exhaustive check of fft_engine.h's lds_delta: for every radix plan of the library (parsed from the sources: the Plan<N> specialisations of
fft_engine.h and the OCEAN_R(...) plans of ocean_kernels.h's geometry table), every stage, every butterfly j and every leg i, the padded LDS
index of the accessed element equals the padded index of the butterfly's first element plus a compile-time constant (reads: j + i * N/R;
writes: j0 + i * NS).  Prints the number of violations (0); tests/test_host_logic.py runs it on every CPU run."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "watersurfacerendering_amd", "csrc")


def plans_from_sources():
    plans = {}
    eng = open(os.path.join(CSRC, "fft_engine.h")).read()
    for n, radices in re.findall(r"struct Plan<(\d+)>\s*:\s*Radices<([\d,\s]+)>", eng):
        plans.setdefault(int(n), []).append([int(r) for r in radices.split(",")])
    ker = open(os.path.join(CSRC, "ocean_kernels.h")).read()
    for n, args in re.findall(r"^OCEAN_GEO\((\d+),(.*)\)\s*$", ker, flags=re.M):
        for radices in re.findall(r"OCEAN_R\(([\d,\s]+)\)", args):
            plans.setdefault(int(n), []).append([int(r) for r in radices.split(",")])
    return plans


def pad(x):
    return x + x // 16


def violations(plans):
    bad = 0
    for n, pls in plans.items():
        for plan in pls:
            prod = 1
            for r in plan:
                prod *= r
            assert prod == n, (n, plan)
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
    return bad


if __name__ == "__main__":
    p = plans_from_sources()
    print("plans:", {n: v for n, v in sorted(p.items())})
    print("violations:", violations(p))
