"""Host-side invariants of the kernels' index arithmetic, checked without a GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lds_offsets_are_compile_time_constants_for_every_plan_in_the_sources():
    """fft_engine.h addresses the R exchange accesses of a butterfly as ONE base address plus lds_delta(d), a compile-time constant:
    valid only while the padded index (idx + idx/16) of idx0 + d splits into that of idx0 plus that of d for every access of every
    stage.  Checked exhaustively for every radix plan the sources define (a new plan is picked up by the parser)."""
    t = _tool("check_lds_offsets")
    plans = t.plans_from_sources()
    assert set(plans) == {16, 32, 64, 128, 256, 512, 1024, 2048, 4096}
    assert [8, 8, 8, 4] in plans[2048] and [8, 8, 8, 8] in plans[4096] and [8, 8, 4, 4] in plans[1024]     # the z-pass plans are seen
    assert t.violations(plans) == 0
    assert t.violations({48: [[3, 16]]}) > 0          # the checker can fail: with a radix that is no power of two the padding term does not split
