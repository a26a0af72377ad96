"""The trace-analysis helpers under bench/ on a synthetic rocprofv3 kernel trace (CPU only)."""
import csv
import gzip
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _trace(path, gz=False):
    rows = []
    t = 1_000_000

    def k(name, dur_us, queue=1, gap_us=0.0, grid=256):
        nonlocal t
        t += int(gap_us * 1000)
        rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + int(dur_us * 1000),
                     "Queue_Id": queue, "Grid_Size_X": grid, "Workgroup_Size_X": 256})
        t += int(dur_us * 1000)

    for step in range(3):
        k("void ddl::(anonymous namespace)::stem_conv_fwd_kernel(ddl::X)", 100.0, gap_us=50.0)
        k("void ddl::(anonymous namespace)::conv_gemm_kernel<64, 128>(ddl::ConvGemmParams)", 80.0)
        k("void ddl::(anonymous namespace)::wgrad_group_kernel<9>(ddl::WgradGroupParams)", 40.0, queue=4, gap_us=9.0)
        k("void ddl::(anonymous namespace)::adam_flat_kernel(float*)", 30.0, gap_us=1.0)
    opener = gzip.open if gz else open
    with opener(path, "wt", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    return path


def _run(script, *args):
    r = subprocess.run([sys.executable, str(ROOT / "bench" / script), *map(str, args)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_gap_report_finds_the_boundary_and_the_fork(tmp_path):
    out = _run("gap_report.py", _trace(tmp_path / "t_kernel_trace.csv.gz", gz=True), "--min", "6")
    assert "last step: 4 kernels" in out
    assert "idle   50.0 us" in out and "before stem_conv_fwd_kernel" in out
    assert "idle    9.0 us" in out and "before wgrad_group_kernel<9> (q4)" in out
    assert "idle in all: 60.0 us" in out


def test_fwd_timeline_lists_the_last_step(tmp_path):
    out = _run("fwd_timeline.py", _trace(tmp_path / "t_kernel_trace.csv"), "--count", "10")
    lines = [l for l in out.splitlines() if " us  +" in l]
    assert len(lines) == 4 and "stem_conv_fwd_kernel" in lines[0] and "q 4" in lines[2]
    assert "conv_gemm_kernel<64, 128>" in lines[1]
