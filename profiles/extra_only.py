#!/usr/bin/env python3
"""Timing of the extraction stage with SV / MOD rows (chr20-30x, ~30 000 MOD + ~400 SV rows, tests/test_scale_gpu.py's case) for builds in
longphase-s_amd/csrc/ab/: python3 profiles/extra_only.py lib1.so lib2.so ...   Each build runs in a child process (LPS_HIP_LIB); no parity check here."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))


def child():
    from lps import abi, hip
    from lps.synth_gpu import SynthGpu
    from lps.synth import make_extras_fast
    CHR20_30X = dict(seed=101, contig_len=64_444_167, n_snp=60_000, coverage=30.0)       # (tests/test_scale_gpu.py)
    g = SynthGpu(0, **CHR20_30X); h = g.to_host(); g.close()
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0); R = abi.Reads.from_synth(h)
    X = abi.extra_from_arrays(*make_extras_fast(h, seed=7))
    with hip.Context(0, abi.default_params()) as ctx:
        ctx.load_chromosome(V, h.ref, R)
        ctx.L.lps_set_stage_timing(ctx.h, 2)
        res = {}
        for tag, x in (("with", X), ("without", None)):
            ctx.set_extra(x); ex, tot, st = [], [], {}
            for _ in range(8):
                ctx.run_phase(); tm = ctx.timings(); ex.append(tm["stages"]["extract"]); tot.append(tm["ms_total"])
                for k, v in tm["stages"].items():
                    st[k] = min(st.get(k, 1e9), v)
            res[tag] = (min(ex), min(tot), st)
        if os.environ.get("LPS_STAGES"):
            print(" | ".join(f"{k} {res['without'][2][k]:.3f}->{res['with'][2][k]:.3f}" for k in res["with"][2] if res["with"][2][k] >= 0.004), flush=True)
        print(f"{os.path.basename(os.environ.get('LPS_HIP_LIB', 'in-tree'))}: extract stage {res['without'][0]:.3f} -> {res['with'][0]:.3f} ms, "
              f"whole step (every stage timed) {res['without'][1]:.3f} -> {res['with'][1]:.3f} ms = {res['with'][1] / res['without'][1]:.3f}x", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for lib in sys.argv[1:] or [""]:
            env = dict(os.environ)
            if lib:
                env["LPS_HIP_LIB"] = os.path.join(ROOT, "longphase-s_amd", "csrc", "ab", lib)
            subprocess.run([sys.executable, __file__, "--child"], env=env, timeout=400)
