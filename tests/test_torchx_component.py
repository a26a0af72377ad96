import pytest


def test_component_builds_appdef_or_command():
    from torchx_component.submit_single import python, launch_command

    with pytest.raises(ValueError):
        python(script="a.py", m="mod")                    # exactly one of -m/-c/--script
    app = python("--epochs", "1", script="single.py", name="single", cpu=4, gpu=1, memMB=16384,
                 mounts=["type=volume,src=nas-pvc,dst=/mnt/dcornelius"])
    role = app.roles[0]
    assert role.entrypoint == "python" and role.args[:1] == ["single.py"] and "--epochs" in role.args
    assert role.num_replicas == 1 and role.resource.gpu == 1 and role.resource.memMB == 16384
    assert role.mounts and role.mounts[0].dst_path == "/mnt/dcornelius"
    cmd = launch_command("ddp.py", nnodes=1, nproc_per_node=8, script_args=["--epochs", "1"])
    assert cmd[:3] == ["python", "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[-3:] == ["ddp.py", "--epochs", "1"]
