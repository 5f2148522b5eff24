#!/usr/bin/env python3
"""Timing experiments on the extraction kernel alone: python3 profiles/extract_only.py [--workload chr1_50x] [--n 8] lib1.so lib2.so ...
Each library (a build in longphase-s_amd/csrc/ab/) runs in a child process with LPS_EXTRACT_ONLY=1: lps_phase_chromosome stops after the
extraction (rc 77) and reports the stage's hipEvent time.  For builds that leave work out on purpose (ablations): their later stages never run."""
import argparse, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "longphase-s_amd"))


def child(a):
    sys.path.insert(0, ROOT)
    import bench
    from lps import abi, hip
    from lps.synth_gpu import SynthGpu
    spec = bench.workload_contigs(a.workload, 201)[0]
    g = SynthGpu(0, **{k: v for k, v in spec.items() if k != "name"})
    V = g.variants(); ctx = hip.Context(0, abi.default_params())
    ctx.load_chromosome_device(V, g.host("ref"), g.device_batch(), g.n_reads)
    ctx.set_stage_timing(1)
    out = abi.PhaseOut(V.n); ms = []
    for _ in range(a.n + 2):
        rc = ctx.L.lps_phase_chromosome(ctx.h, C.byref(out.c))
        assert rc == 77, rc
        t = abi.Timings(); ctx.L.lps_get_timings(ctx.h, C.byref(t)); ms.append(t.ms_kernel[1])
    assert ctx.L.lps_stage_name(1).decode() == "extract"
    ms = sorted(ms[2:])
    print(f"{os.path.basename(os.environ.get('LPS_HIP_LIB', 'in-tree'))}: extract min {ms[0]:.4f} median {ms[len(ms) // 2]:.4f} ms", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="chr1_50x"); ap.add_argument("--n", type=int, default=8); ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child:
        child(a)
    else:
        for rnd in range(2):
            for lib in a.libs:
                env = dict(os.environ, LPS_EXTRACT_ONLY="1", LPS_HIP_LIB=os.path.join(ROOT, "longphase-s_amd", "csrc", "ab", lib))
                subprocess.run([sys.executable, __file__, "--child", "--workload", a.workload, "--n", str(a.n)], env=env, timeout=300)
