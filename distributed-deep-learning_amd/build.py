"""Build the gfx950 HIP extension in-tree: ``csrc/*.hip`` + ``csrc/bindings.cpp`` -> ``csrc/_ddl_hip.so``.

Kernels are plain HIP translation units with C launchers (raw pointers + ``hipStream_t``, no torch
headers -> seconds per file); only ``bindings.cpp`` sees torch/pybind11.  Objects are compiled
with ``hipcc --offload-arch=gfx950`` (cross-compiles without a GPU) and linked against the HIP
runtime that PyTorch itself ships (``torch/lib/libamdhip64.so``) so that exactly one runtime is
loaded in the process.  The ``.so`` stays in the source tree: it travels to the GPU box with the
repo snapshot and shows up as loaded native code there.

    python -m ddl.build            # incremental
    python -m ddl.build --force
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import List

ARCH = "gfx950"
CSRC = Path(__file__).resolve().parent / "csrc"
OUT = CSRC / "_ddl_hip.so"
OBJ_DIR = CSRC / "build"


def _torch_paths():
    import torch
    from torch.utils import cpp_extension as ce

    tdir = Path(torch.__file__).resolve().parent
    return [str(p) for p in ce.include_paths("cuda")], str(tdir / "lib")


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    if r.stdout.strip() and os.environ.get("DDL_BUILD_VERBOSE"):
        print(r.stdout)


def _stamp(src: Path, flags: List[str]) -> str:
    h = hashlib.sha1()
    h.update(src.read_bytes())
    for hdr in sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hpp")):
        h.update(hdr.read_bytes())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    incs, tlib = _torch_paths()
    OBJ_DIR.mkdir(exist_ok=True)
    hip_flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}",
                 "-fno-gpu-rdc", "-Wno-unused-result", "-DNDEBUG", f"-I{CSRC}"]
    cpp_flags = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1",
                 "-DUSE_ROCM=1", "-DHIPBLAS_V2", "-DTORCH_EXTENSION_NAME=_ddl_hip",
                 "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=1",
                 "-Wno-unused-result", "-Wno-deprecated-declarations",
                 f"-I{CSRC}", f"-I{sysconfig.get_paths()['include']}"] + [f"-I{i}" for i in incs]
    extra = os.environ.get("DDL_BUILD_DEFINES", "").split()      # e.g. -DDDL_PHASE_PROBE (bench/phase_probe.py)
    hip_flags += extra + os.environ.get("DDL_HIP_FLAGS", "").split()   # extra device-compiler flags (experiments)
    cpp_flags += extra
    jobs = []
    for src in sorted(CSRC.glob("*.hip")):
        jobs.append((src, hip_flags))
    for src in sorted(CSRC.glob("*.cpp")):
        jobs.append((src, cpp_flags))
    objs, todo = [], []
    for src, flags in jobs:
        obj = OBJ_DIR / (src.name + ".o")
        stamp_file = OBJ_DIR / (src.name + ".stamp")
        stamp = _stamp(src, flags)
        objs.append(obj)
        if force or not obj.exists() or not stamp_file.exists() or stamp_file.read_text() != stamp:
            todo.append((src, flags, obj, stamp_file, stamp))

    def compile_one(job):
        src, flags, obj, stamp_file, stamp = job
        # .cpp = host-only binding code: plain clang++ (hipcc would add a useless device pass)
        cc = hipcc if src.suffix == ".hip" else os.environ.get("DDL_CXX", "/opt/rocm/lib/llvm/bin/clang++")
        _run([cc, *flags, "-c", str(src), "-o", str(obj)])
        stamp_file.write_text(stamp)
        if verbose:
            print(f"[ddl.build] compiled {src.name}", flush=True)

    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(compile_one, todo))
    if todo or not OUT.exists() or force:
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", str(OUT),
              *[str(o) for o in objs], f"-L{tlib}", f"-Wl,-rpath,{tlib}",
              "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
              "-lrccl",        # the RCCL that PyTorch itself loads (torch/lib/librccl.so): csrc/rccl_comm.cpp
              "-lz"])          # zlib: csrc/png_loader.cpp
        if verbose:
            print(f"[ddl.build] linked {OUT}", flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
