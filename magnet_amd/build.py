"""Build libmagnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m magnet_amd.build [--force] [--dev]

Flags that matter: -ffp-contract=off (the kernels mirror the reference's separately-rounded
multiply/add; fused operations are written as explicit fmaf), correctly-rounded fp32 divide
(hipcc's default, stated explicitly)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmagnet_hip.so")
DEV_LIB = os.path.join(HERE, "libmagnet_hip_dev.so")      # -DMAGNET_DEV build, loaded only by tools/ (lib.use_dev_build())
SOURCES = ["api.hip", "cost_volume.hip", "cost_volume_worklist.hip", "cost_volume_cand.hip", "cost_volume_fast.hip", "cost_volume_fast64.hip", "cost_volume_v3.hip", "cost_volume_f_bwd.hip", "cost_volume_f_gather.hip", "conv_mfma.hip", "fnet_kernels.hip", "elementwise.hip"]
DEV_SOURCES = ["cost_volume_v4.hip", "cost_volume_v5.hip"]     # round 4's measured-and-lost matcher experiments: records, compiled into the dev library only
HEADERS = ["cv_common.hpp", "cv_fast_common.hpp", "cv_runs.hpp", "conv_common.hpp", "warp_math.hpp", os.path.join("..", "..", "include", "magnet_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fvisibility=hidden",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + (DEV_SOURCES if lib == DEV_LIB else []) + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


# Per-file flags.  The matcher kernels are VALU-issue-bound; clang's SLP vectoriser packs adjacent fp32 mul/add/fma into
# v_pk_*_f32, which on gfx950 run at HALF the per-instruction rate of their scalar forms (tools/ubench/valu_rate.hip: 4.9 vs
# 2.5 cycles) and need their operands copied into aligned register pairs (55 v_mov per 4 views): packing is a net loss there.
EXTRA_FLAGS = {"cost_volume_fast.hip": ["-fno-slp-vectorize"], "cost_volume_fast64.hip": ["-fno-slp-vectorize"],
               "cost_volume_v3.hip": ["-fno-slp-vectorize"], "cost_volume_v4.hip": ["-fno-slp-vectorize"], "cost_volume_v5.hip": ["-fno-slp-vectorize"]}


def build(force: bool = False, verbose: bool = False, dev: bool | None = None) -> str:
    """dev=True: a SECOND library, libmagnet_hip_dev.so, compiled with -DMAGNET_DEV: it honours
    MagnetCostVolumeArgs.dev_flags and the MAGNET_* environment switches of tools/ (kernel variants, timing ablations).
    The product library ignores all of them and is never replaced by this build; only tools/ load the dev library."""
    dev = bool(dev)
    if not force and not _stale(DEV_LIB if dev else LIB):
        return DEV_LIB if dev else LIB
    objs = []
    procs = []
    for s in SOURCES + (DEV_SOURCES if dev else []):
        o = os.path.join(CSRC, s.replace(".hip", ".dev.o" if dev else ".o"))
        cmd = [hipcc(), *FLAGS, *(["-DMAGNET_DEV"] if dev else []), *EXTRA_FLAGS.get(s, []), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    out = DEV_LIB if dev else LIB
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
