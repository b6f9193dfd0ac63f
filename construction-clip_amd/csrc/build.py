"""Build libcclip_hip.so (gfx950 only) in-tree: hipcc each .hip to an object, link one shared library.

The built .so lands in construction-clip_amd/cclip_hip/ (git-ignored, but it travels to the GPU
box with the tree).  No JIT cache, no torch extension machinery: the library has a plain C ABI
(include/cclip_hip.h) and is loaded with ctypes.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "cclip_hip")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libcclip_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "cclip_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


# translation units that touch 16-bit operands are built twice: bf16 (default) and IEEE fp16 (-DCCLIP_F16)
DUAL = ("gemm_bf16", "attention", "layernorm", "embed", "loss", "optim", "decode")   # "attention" also matches attention_small


def _jobs():
    jobs = []
    for src in _sources():
        jobs.append((src, False))
        if src.startswith(DUAL):
            jobs.append((src, True))
    return jobs


def _compile(job, force: bool) -> str:
    src, f16 = job
    obj = os.path.join(OBJ_DIR, src[:-4] + ("_f16" if f16 else "") + ".o")
    spath = os.path.join(HERE, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, *(["-DCCLIP_F16"] if f16 else []), "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _jobs()
    with ThreadPoolExecutor(max_workers=min(7, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[cclip_hip] built {LIB} from {len(srcs)} objects")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
