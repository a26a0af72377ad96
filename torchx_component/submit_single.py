"""torchx-style launch components for the MI355X harness.

Reference: torchx_component/submit_single.py:37-78 - a ``python(...)`` component (stock
``torchx.components.utils.python`` plus ``mounts``) used as
``torchx run -s kubernetes ./component/single.py:python --script single.py --mount ... --gpu 1``
(command:2); the distributed scripts go through the stock ``dist.ddp -j NxM`` component
(command:5-34).

torchx is not installed in this image, so it is imported lazily: with torchx present
``python()`` / ``ddp()`` return real ``torchx.specs.AppDef`` objects (usable with ``torchx run``);
without it they return structurally identical light-weight stand-ins, and ``launch_command`` /
``main`` turn a job description into the equivalent ``torchrun`` command line for one
8xMI355X node (``python -m torch.distributed.run --nproc-per-node N script.py ...``).
"""
from __future__ import annotations

import os
import shlex
import subprocess
import sys
from dataclasses import dataclass, field
from typing import Dict, List, Optional

DEFAULT_IMAGE = "rocm/pytorch:rocm7.0_ubuntu22.04_py3.10_pytorch_2.10"

try:                                                    # pragma: no cover - torchx absent in this image
    import torchx
    from torchx import specs
    _HAVE_TORCHX = True
    DEFAULT_IMAGE = getattr(torchx, "IMAGE", DEFAULT_IMAGE)
except Exception:                                        # noqa: BLE001
    _HAVE_TORCHX = False

    @dataclass
    class _Resource:
        cpu: int = 1
        gpu: int = 0
        memMB: int = 1024
        capabilities: Dict[str, str] = field(default_factory=dict)

    @dataclass
    class _Mount:
        src: str
        dst_path: str
        read_only: bool = False
        kind: str = "volume"

    @dataclass
    class _Role:
        name: str
        image: str
        entrypoint: str
        args: List[str]
        env: Dict[str, str]
        num_replicas: int
        resource: _Resource
        mounts: List[_Mount]

    @dataclass
    class _AppDef:
        name: str
        roles: List[_Role]

    class specs:                                         # minimal stand-in for torchx.specs
        AppDef, Role = _AppDef, _Role

        @staticmethod
        def resource(cpu=None, gpu=None, memMB=None, h=None):
            return _Resource(cpu or 1, gpu or 0, memMB or 1024, {"named": h} if h else {})

        @staticmethod
        def parse_mounts(opts: List[str]) -> List[_Mount]:
            """``type=volume,src=nas-pvc,dst=/mnt/x[,readonly]`` (torchx mount syntax, command:2)."""
            out = []
            for o in opts:
                kv = dict(p.split("=", 1) if "=" in p else (p, "true") for p in o.split(","))
                out.append(_Mount(kv.get("src", ""), kv.get("dst", ""), "readonly" in kv,
                                  kv.get("type", "volume")))
            return out


def python(*args: str, m: Optional[str] = None, c: Optional[str] = None, script: Optional[str] = None,
           image: str = DEFAULT_IMAGE, name: str = "single", cpu: int = 1, gpu: int = 0,
           memMB: int = 1024, h: Optional[str] = None, num_replicas: int = 1,
           mounts: Optional[List[str]] = None):
    """Single-role ``python`` app (reference component, used for single.py)."""
    if sum([m is not None, c is not None, script is not None]) != 1:
        raise ValueError("exactly one of `-m`, `-c` and `--script` needs to be specified")
    cmd = [script] if script else (["-m", m] if m else ["-c", c])
    return specs.AppDef(
        name=name,
        roles=[specs.Role(name="python", image=image, entrypoint="python", num_replicas=num_replicas,
                          resource=specs.resource(cpu=cpu, gpu=gpu, memMB=memMB, h=h),
                          args=[*cmd, *args], env={"HYDRA_MAIN_MODULE": m} if m else {},
                          mounts=specs.parse_mounts(mounts) if mounts else [])],
    )


def launch_command(script: str, nnodes: int = 1, nproc_per_node: int = 1, script_args: Optional[List[str]] = None,
                   master_addr: str = "127.0.0.1", master_port: int = 29500) -> List[str]:
    """The torchrun command a ``dist.ddp -j {nnodes}x{nproc}`` job runs in each replica."""
    return ["python", "-m", "torch.distributed.run", f"--nnodes={nnodes}", f"--nproc-per-node={nproc_per_node}",
            "--master-addr", master_addr, "--master-port", str(master_port), script, *(script_args or [])]


def ddp(*script_args: str, script: str, j: str = "1x1", image: str = DEFAULT_IMAGE, name: Optional[str] = None,
        cpu: int = 4, gpu: int = 1, memMB: int = 16384, h: Optional[str] = None,
        mounts: Optional[List[str]] = None, env: Optional[Dict[str, str]] = None):
    """``dist.ddp``-shaped component: ``-j {nnodes}x{nproc_per_node}`` replicas each running torchrun
    (reference: command:5-34).  On MI355X one node is ``-j 1x8``."""
    nnodes, nproc = (int(v) for v in j.split("x"))
    stem = os.path.splitext(os.path.basename(script))[0]
    cmd = launch_command(script, nnodes, nproc, list(script_args))
    e = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", **(env or {})}
    return specs.AppDef(
        name=name or stem,
        roles=[specs.Role(name=stem, image=image, entrypoint=cmd[0], args=cmd[1:], env=e, num_replicas=nnodes,
                          resource=specs.resource(cpu=cpu, gpu=gpu, memMB=memMB, h=h),
                          mounts=specs.parse_mounts(mounts) if mounts else [])],
    )


def run_local(app) -> int:
    """Run an AppDef on this node (the local scheduler): one process per role replica."""
    rc = 0
    for role in app.roles:
        env = dict(os.environ, **role.env)
        env.setdefault("TORCHX_JOB_ID", f"local://{app.name}/{app.name}-{os.getpid()}")
        rc |= subprocess.call([role.entrypoint if role.entrypoint != "python" else sys.executable, *role.args], env=env)
    return rc


def main(argv: Optional[List[str]] = None) -> int:
    import argparse

    p = argparse.ArgumentParser(description="submit a ddl job on this node (torchx-style)")
    p.add_argument("component", choices=["python", "ddp"])
    p.add_argument("--script", required=True)
    p.add_argument("-j", default="1x1")
    p.add_argument("--dry-run", action="store_true")
    a, rest = p.parse_known_args(argv)
    app = python(*rest, script=a.script) if a.component == "python" else ddp(*rest, script=a.script, j=a.j)
    if a.dry_run:
        for r in app.roles:
            print(shlex.join([r.entrypoint, *r.args]))
        return 0
    return run_local(app)


if __name__ == "__main__":
    sys.exit(main())
