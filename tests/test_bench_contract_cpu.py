"""bench.py contract (CPU-checkable parts): the algorithmic-work model of SURVEY.md §8d D3, the decode FLOP model, and
the shape of the JSON line (checked on the committed evidence file produced by `python bench.py` on the GPU box)."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench                      # noqa: E402
import bench_decode               # noqa: E402


def test_algorithmic_flops_match_survey_d3():
    # C3 = (24.9 G * 200 + 15.4 M * (200*81 - 1640)) * 50 = 260.2 TFLOP; C2 = 130.9 TFLOP
    assert bench.algorithmic_flops_per_forward(200, 64) * 50 == pytest.approx(260.2e12, rel=5e-3)
    assert bench.algorithmic_flops_per_forward(400, 32) * 50 == pytest.approx(130.9e12, rel=5e-3)
    # short clips attend to every frame: b * T^2
    assert bench.algorithmic_flops_per_forward(16, 32) == pytest.approx(6.25e9 * 16 + 3.85e6 * 256, rel=1e-2)


def test_decode_flop_model_counts_the_per_frame_convolutions():
    res, be = 256, 64
    hb, cb = res // 4, 4 * be
    f = 12 * 2 * hb * hb * 9 * cb * cb                     # 6 ResBlock2d x 2 convs, 256 -> 256 at 64^2
    f += 2 * (2 * hb) ** 2 * 9 * cb * (cb // 2)            # up block 0: 256 -> 128 at 128^2
    f += 2 * res * res * 9 * (cb // 2) * be                # up block 1: 128 -> 64 at 256^2
    f += 2 * res * res * 49 * be * 3                       # final 7x7, 64 -> 3
    assert bench_decode.decode_flops_per_frame(res) == pytest.approx(f, rel=1e-12)
    assert bench_decode.decode_flops_per_frame(res) * 200 == pytest.approx(15.7e12, rel=1e-2)


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r1_final_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    frames = d["config"]["frames_per_gpu"]
    assert d["value"] == pytest.approx(frames / (d["ms_per_step"] * 1e-3), rel=1e-6)      # whole-job frames / timed seconds
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    # the flow-decode report is beside the metric, never inside it
    assert "flow_decode" in d and d["flow_decode"]["sampler_plus_decode_frames_per_s"] < d["value"]


def test_bench_defaults_are_the_headline_config():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for frag in ('"--gpus", type=int, default=1', '"--frames", type=int, default=200', '"--res", type=int, default=256',
                 '"--ddim-steps", type=int, default=50'):
        assert frag in src, frag


def test_launch_plan_never_downgrades_a_multi_gpu_request(tmp_path):
    """`python bench.py --gpus 8` called plainly must start 8 ranks itself (or fail), never print `n_gpus: 1`."""
    import subprocess
    assert bench.launch_plan(1, {}, 0, []) == ("run", None)
    assert bench.launch_plan(1, {}, 8, ["--steps", "3"]) == ("run", None)
    # started by torch.distributed.run with the matching rank count: this process is a rank
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, 8, []) == ("run", None)
    # plain call with N > 1: spawn N ranks, rendezvous on the loopback interface, arguments passed through
    what, cmd = bench.launch_plan(8, {}, 8, ["--gpus", "8", "--steps", "2", "--mode", "replica"])
    assert what == "spawn" and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "2", "--mode", "replica"] and cmd[-7] == os.path.join(ROOT, "bench.py")
    # requests that cannot be honoured exit with code 2
    for gpus, env, ndev in ((8, {}, 1), (2, {}, 0), (8, {"WORLD_SIZE": "1"}, 8), (2, {"WORLD_SIZE": "4"}, 8), (0, {}, 8)):
        with pytest.raises(SystemExit) as e:
            bench.launch_plan(gpus, env, ndev, [])
        assert e.value.code == 2
    # the command line itself, on this box (no GPU): non-zero exit, nothing on stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == "" and "--gpus 2" in r.stderr
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout.strip() == ""
    # the spawn command really starts N ranks with RANK / WORLD_SIZE set (a stand-in script in place of bench.py)
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys\nopen(os.path.join(sys.argv[1], 'rank' + os.environ['RANK']), 'w').write(os.environ['WORLD_SIZE'])\n")
    what, cmd = bench.launch_plan(2, {}, 2, [str(tmp_path)])
    cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = str(probe)
    assert subprocess.call(cmd, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}) == 0
    assert sorted(p.name for p in tmp_path.glob("rank*")) == ["rank0", "rank1"] and (tmp_path / "rank1").read_text() == "2"


def test_tshard_preflight_failure_is_loud():
    """`--mode tshard` with a failed preflight exits with code 3 unless --allow-fallback was given."""
    assert bench.resolve_mode("tshard", True, False) == "tshard"
    assert bench.resolve_mode("replica", False, False) == "replica"
    assert bench.resolve_mode("tshard", False, True) == "replica"
    with pytest.raises(SystemExit) as e:
        bench.resolve_mode("tshard", False, False)
    assert e.value.code == 3


def test_stdout_redirection_keeps_foreign_prints_off_the_json_stream(tmp_path):
    """RCCL prints its banner with printf on fd 1 when a communicator is created; bench.py's stdout carries one JSON line.  The
    fd-level redirection moves such output to stderr and restores stdout afterwards."""
    import subprocess
    # (the banner is written with the C library's buffered printf, as RCCL does: with stdout a pipe it would otherwise sit in the C
    #  buffer and come out at process exit, AFTER the JSON line -- seen on the GPU box in round 5)
    code = ("import os, sys, ctypes; sys.path.insert(0, %r); import bench\n"
            "libc = ctypes.CDLL(None)\n"
            "with bench.stdout_to_stderr():\n    os.write(1, b'RCCL version : banner\\n'); libc.printf(b'HIP version  : banner2\\n')\n"
            "print('{\"metric\": 1}')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == '{"metric": 1}' and "banner" in r.stderr and "banner2" in r.stderr
