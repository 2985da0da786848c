"""CPU: the parts of bench.py that do not need a GPU - the workload naming (BASELINE.json configs), the algorithmic FLOP
accounting of SURVEY.md 8d and the record builders - so that the driver-facing JSON contract cannot drift silently."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_workload_names_follow_the_arguments():
    assert bench.workload_name(64, 224, 1000, 1).startswith("BASELINE.json configs[1]")
    assert bench.workload_name(64, 224, 70000, 1).startswith("BASELINE.json configs[3]")
    assert bench.workload_name(32, 448, 1000, 1).startswith("BASELINE.json configs[4]")
    assert bench.workload_name(8, 64, 50, 1).startswith("custom")           # the rehearsal size is NOT configs[1]
    assert "10 critic updates" in bench.workload_name(64, 224, 1000, 10)
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert "batch 64" in cfgs[1] and "448" in cfgs[4] and "70k" in cfgs[3]


def test_conv_flops_per_step_matches_the_survey():
    # 8*F_fwd - 2*f1 = 12 489.5 GFLOP at configs[1], 127.4 at configs[0], 24 979 at configs[4]
    assert abs(bench.conv_flops_per_step(64, 224) / 1e9 - 12489.5) < 1.0
    assert abs(bench.conv_flops_per_step(8, 64) / 1e9 - 127.4) < 0.1
    assert abs(bench.conv_flops_per_step(32, 448) / 1e9 - 24979.0) < 2.0


class _Ev:
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return other.t - self.t


def test_roofline_records_from_event_timings():
    timing = [("conv_halo3_kernel<2,128,2,2,true,true,false,false>", 236.8e9, 0.0, _Ev(0.0), _Ev(0.6)),
              ("conv_halo3_kernel<2,128,2,2,true,true,false,false>", 236.8e9, 0.0, _Ev(1.0), _Ev(1.6)),
              ("conv_s2_kernel<false,true,7>", 328.8e9, 0.0, _Ev(2.0), _Ev(2.9)),
              ("adam_kernel", 0.0, 967e6, _Ev(3.0), _Ev(3.16))]
    per = bench.summarise_timing(timing)
    r = bench.conv_roofline(per, 2, dt=0.01)
    assert r["kernel"].startswith("conv_halo3_kernel") and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - 2 * 236.8e9 / 1.2e-3 / 1e12) < 1e-6 and abs(r["peak"] - 2500.0 / 3) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["launches"] == 2
    f32 = bench.conv_roofline(per, 0, dt=0.01)
    assert f32["peak"] == 157.3
    hbm = bench.hbm_rooflines(per, steps=1)
    assert [h["kernel"] for h in hbm] == ["adam_kernel"]
    assert abs(hbm[0]["GBps"] - 967e6 / 0.16e-3 / 1e9) < 1e-6 and abs(hbm[0]["frac_of_8TBps"] - hbm[0]["GBps"] / 8000.0) < 1e-12


def test_exit_code_travels_in_the_json_line():
    """torch.distributed.run collapses a rank's exit code to 1: the parent derives the parity code 3 from rank 0's line."""
    assert bench.exit_code_from(0, json.dumps({"parity": {"ok": True}})) == 0
    assert bench.exit_code_from(1, json.dumps({"parity": {"ok": False}})) == 3
    assert bench.exit_code_from(0, json.dumps({"parity": {"ok": False}})) == 3
    assert bench.exit_code_from(1, None) == 1 and bench.exit_code_from(1, "not json") == 1 and bench.exit_code_from(0, json.dumps({"value": 1})) == 0


def test_cli_contract_flags_exist():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--critic-iters", "--cpu-rows", "--f32-steps", "--serial-steps", "--single-stream",
                 "--other-configs", "--other-oracle-rows", "--bitwise-iters"):
        assert flag in out.stdout


def test_gpus_n_starts_its_own_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher on the command line and no WORLD_SIZE in the environment (the form the driver
    may use): bench.py starts `python -m torch.distributed.run` as a child, the two ranks meet over gloo (SGG_DP_BACKEND; "nccl" =
    RCCL on a GPU node), all-reduce a bucket through sgg_amd.dp.GradReducer, and the parent relays rank 0's ONE JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SGG_DP_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl"]["nranks"] == 2 and rec["rccl"]["backend"] == "gloo" and rec["rccl"]["allreduce_mean_ok"]


def test_self_launch_relays_the_exit_code_of_the_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SGG_DP_BACKEND"] = "gloo"
    # without --rendezvous-only the ranks need MI355X GPUs: on a CPU box they must fail loudly (no CPU fallback), and so must the parent
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                          "--size", "32", "--vocab", "11", "--cpu-rows", "0"], capture_output=True, text=True, timeout=600, env=env)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_line_names_its_schedule_and_measures_the_roofline_in_serial_steps():
    """The headline is the product's default (two-stream) schedule and says so; `roofline` comes from the serial leg, whose dominant
    kernel's launches per step times their average duration must fit inside a serial step (a rehearsal-sized workload)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--size", "64",
                          "--vocab", "50", "--cpu-rows", "4", "--f32-steps", "1", "--ci10-steps", "0", "--serial-steps", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["config"]["schedule"] == "two-stream; per-kernel figures from serial steps"
    assert rec["config"]["workload"].startswith("custom") and rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["serial"]["steps"] == 2 and rec["serial"]["ms_per_step"] > 0 and "other_configs" not in rec      # (not the default workload)
    r = rec["roofline"]
    assert r["bound"] == "mfma" and r["launches_per_step"] * r["avg_launch_ms"] <= rec["serial"]["ms_per_step"]
    assert any(h["kernel"].startswith("conv_c3") for h in rec["roofline_hbm"]), [h["kernel"] for h in rec["roofline_hbm"]]
    assert rec["parity"]["ok"] is True and rec["cpu_baseline"]["kind"] == "port" and rec["native_f32"]["steps"] == 1
    # the oracle record is about the schedule that was timed, and that schedule is bit-identical to the serial one at the timed size
    sch = rec["parity"]["precision2_vs_oracle"]["schedule"]
    assert sch["overlap_streams"] and sch["head_side_stream"] and sch["g_early_fired"], sch
    assert rec["parity"]["two_stream_bitwise_at_full_size"] is True and rec["parity"]["two_stream_bitwise_detail"]["g_early_fired"]
    assert rec["parity"]["precision2_vs_oracle"]["margin_over_logit_tolerance"] > 4.0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "4", "--size", "64",
                          "--vocab", "50", "--cpu-rows", "0", "--f32-steps", "0", "--ci10-steps", "0", "--serial-steps", "0", "--single-stream"],
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    rec1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert rec1["config"]["schedule"].startswith("serial") and "roofline" not in rec1
