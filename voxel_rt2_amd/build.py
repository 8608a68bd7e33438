"""Build libvrt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m voxel_rt2_amd.build [--force]

Flags that matter for the numeric contract (include/vrt_detmath.h): -ffp-contract=off so no
a*b+c is fused behind our back, no fast-math, correctly rounded f32 divide / sqrt (hipcc default,
stated explicitly), denormals kept.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvrt_hip.so")
SOURCES = ["vrt_kernels.hip", "vrt_sky_kernels.hip", "vrt_api.hip"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-gpu-flush-denormals-to-zero",
    # no v_pk_{mul,add,fma}_f32: the two-lane forms need their operands in aligned register pairs, and the moves and the register
    # pressure that buys cost more than the halved issue count saves -- the render kernels spill 16-17 registers with them and
    # none without (same IEEE results per component; config 2 +6 %, sun-lit +6 %, config 3 +4 %, dense 4K +1.5 %).  The host
    # pass ignores the feature name with a warning.
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result",
]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libvrt_hip.so cannot be built (there is no CPU fallback)")


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    inc = os.path.join(os.path.dirname(HERE), "include")
    d += [os.path.join(inc, f) for f in os.listdir(inc)]
    d.append(os.path.abspath(__file__))
    return d


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in _deps())


def source_id():
    """sha256 (first 16 hex digits) over the sources and flags the library is built from: vrt_build_id()."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))
    files += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=False, extra_flags=(), out=None):
    """out: an A/B variant (build_variants/libvrt_<name>.so, loaded through VRT_LIB_PATH) instead of the shipped library."""
    if out is None and not force and not is_stale():
        return OUT
    out = out or OUT
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tag = source_id() + ("+" + "".join(extra_flags) if extra_flags else "")
    cmd = [_hipcc()] + FLAGS + list(extra_flags) + [f'-DVRT_BUILD_ID="{tag}"'] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libvrt_hip.so")
    if verbose and r.stderr:
        sys.stderr.write(r.stderr)
    return out


if __name__ == "__main__":
    # python -m voxel_rt2_amd.build [--force]                      the shipped library
    # python -m voxel_rt2_amd.build --variant NAME -DFLAG ...      build_variants/libvrt_NAME.so with extra flags (A/B runs)
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        name, extra = sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-")]
        print(build(force=True, extra_flags=extra, out=os.path.join(os.path.dirname(HERE), "build_variants", f"libvrt_{name}.so")))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
