"""CPU tier: bench.py's multi-rank harness, end to end, without a GPU. QNNP_BENCH_STUB=1 replaces the device step by a
sleep (rank r is 1 + r/2 times slower than rank 0) and RCCL by gloo; everything else is the code the driver's SCALE run
goes through: `python bench.py --gpus N` re-launching itself under torch.distributed.run on 127.0.0.1, the WORLD_SIZE
check, the barrier-bracketed timed region, the max-over-ranks time, the batch shards of the sweep and the one JSON line
(whole-job value, per-rank launch times, a cpu_baseline object at N > 1 as well)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = dict(os.environ)
    env["QNNP_BENCH_STUB"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [1, 2, 3, 8])
def test_bench_spawns_its_ranks_and_prints_one_line(gpus, tmp_path):
    full_out = str(tmp_path / "bench_full.json")
    res = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--steps", "6", "--warmup", "1", "--sweep-batch", "10",
                          "--full-out", full_out],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _line(res.stdout)
    # the contract line is the LAST line of stdout, short enough for the driver's capture, and carries no `extra`;
    # the per-layer tables are in the --full-out file, which repeats the line's fields
    assert res.stdout.rstrip().splitlines()[-1].startswith("{")
    assert len(res.stdout.rstrip().splitlines()[-1]) < 6000 and "extra" not in line
    full = json.load(open(full_out))
    assert {k: v for k, v in full.items() if k != "extra"} == line
    line["extra"] = full["extra"]
    assert line["n_gpus"] == gpus and line["steps"] == 6 and line["warmup"] == 1
    assert line["metric"] == "q8gemm_int8_tops" and line["unit"] == "TOPS" and line["scaling"] == "weak"
    assert line["data"].startswith("stub")
    # the slowest rank (the last one: 2 ms x (1 + (N-1)/2) per step) sets the job time
    slowest = 2.0 * (1.0 + 0.5 * (gpus - 1))
    assert slowest * 0.95 <= line["ms_per_step"] <= slowest * 3.0, line["ms_per_step"]      # (upper bound: sanity only -- a busy host stretches sleeps)
    # whole-job value: N replicas of the GEMM per step of the slowest rank
    expected = gpus * 2.0 * 4096 ** 3 / (line["ms_per_step"] * 1e-3) / 1e12
    assert abs(line["value"] - expected) <= 0.01 * expected
    assert line["cpu_baseline"] is not None                     # N > 1 lines carry the baseline object too
    sweep = line["extra"]["mobilenetv2_sweep"]
    assert sweep["batch_per_gpu"] == 10 and sweep["shard_start"] == 0       # rank 0's shard of 10 x N images
    # the whole job's sweep figures: N shards in the slowest rank's time
    assert abs(sweep["images_per_s"] - 10 * gpus / (sweep["ms_per_batch"] * 1e-3)) <= 0.01 * sweep["images_per_s"]
    assert abs(sweep["aggregate_hbm_gbs"] - gpus * sweep["aggregate_frac_of_hbm_peak"] * 8000.0) <= 0.02 * sweep["aggregate_hbm_gbs"] + 0.1
    # BASELINE's other configs travel inside `roofline` (the driver's record drops `extra`)
    secondary = line["roofline"]["secondary"]
    assert secondary["c4_sweep_images_per_s_graph"] == sweep["images_per_s"] and secondary["c4_batch_per_gpu"] == 10
    assert all(not isinstance(v, (dict, list)) for v in secondary.values())
    if gpus > 1:
        per_rank = line["roofline"]["per_rank_launch_ms"]
        assert len(per_rank) == gpus and per_rank == sorted(per_rank)       # rank r sleeps longer than rank r-1
        assert "one replica per GPU" in line["config"]["workload"]


def test_launcher_world_size_mismatch_is_refused():
    """started by a launcher with a world size other than --gpus: never report an M-rank run as an N-GPU one"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "2", "--warmup", "0"],
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0
    assert "refusing" in (res.stderr + res.stdout)
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_fewer_gpus_than_ranks_is_refused_without_the_stub():
    """the real path on this GPU-less box: --gpus 2 must fail loudly, not fall back to fewer ranks or to the CPU"""
    env = _env()
    env.pop("QNNP_BENCH_STUB")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the refusal cannot be provoked")
    assert res.returncode != 0
    assert "refusing" in res.stderr or "needs" in res.stderr or "GPU" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_assemble_line_contract_fields():
    sys.path.insert(0, ROOT)
    import bench
    line = bench.assemble_line(world=8, steps=20, warmup=5, ms_per_step=0.07, ev_ms_per_rank=[0.066 + 0.001 * r for r in range(8)],
                               gemm_kernel="q8_gemm_mfma_256x256", info={"arch": "gfx950", "compute_units": 256},
                               roofline={"bound": "mfma", "achieved": 2080.0, "peak": 5033.2, "unit": "TOP/s", "frac": 0.41,
                                         "traffic": 1}, cpu={"value": 1.1, "unit": "TOPS", "cores": 16, "kind": "reference",
                                                             "sample": "x"}, extra={}, data="synthetic")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 8 and line["vs_baseline"] is None and line["dtype"] == "u8"
    assert abs(line["value"] - 8 * 2 * 4096 ** 3 / 0.07e-3 / 1e12) < 1.0
    assert len(line["roofline"]["per_rank_launch_ms"]) == 8 and len(line["roofline"]["per_rank_frac"]) == 8
    assert "model" not in line["config"]


def test_contract_line_stays_short_with_a_realistic_extra(capsys, tmp_path):
    """Round 5's line was 23 KB (per-layer arrays under `extra`) and the driver, which keeps the last 8 KB of stdout, could
    not parse it. The fixture is that very record: emitted through bench.emit it must come out under 6000 bytes, as the
    last stdout line, with roofline / secondary / cpu_baseline intact and every per-layer table in the side file."""
    sys.path.insert(0, ROOT)
    import bench
    record = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_r05f.json")))
    assert len(json.dumps(record)) > 20000 and "layers" in record["extra"]["mobilenetv2_sweep"]
    full_out = str(tmp_path / "sub" / "bench_full.json")
    text = bench.emit(record, full_out)
    out = capsys.readouterr().out
    assert out.rstrip().splitlines()[-1] == text
    assert len(text) < 6000, len(text)
    line = json.loads(text)
    assert "extra" not in line
    assert line["roofline"]["frac"] == record["roofline"]["frac"]
    assert line["roofline"]["secondary"]["c4_sweep_images_per_s_graph"] == record["roofline"]["secondary"]["c4_sweep_images_per_s_graph"]
    assert line["cpu_baseline"]["kind"] == "reference"
    assert json.load(open(full_out)) == record
    # a line that still outgrows the budget sheds its optional blocks instead of being cut by the capture
    fat = json.loads(json.dumps(record))
    fat["roofline"]["secondary"] = {f"k{i}": "x" * 40 for i in range(200)}
    slim = json.loads(bench.emit(fat, ""))
    assert "secondary" not in slim["roofline"] and slim["roofline"]["dropped_for_length"] == ["secondary"]
    assert slim["roofline"]["frac"] == record["roofline"]["frac"] and slim["cpu_baseline"] is not None


def test_cpu_baseline_sample_strings_are_short():
    """`cpu_baseline.sample` says what was timed in < 120 characters (round-5 review): the templates at their widest fill"""
    src = open(BENCH).read()
    assert 'f"reference SSE2 q8gemm, M={M} rows of N=K=4096 x {iters} runs, {best_threads} of {cores} "' in src
    assert len(f"reference SSE2 q8gemm, M={512} rows of N=K=4096 x {400} runs, {128} of {256} threads (best), {12.0:.1f} s") < 120


def test_more_ranks_than_gpus_is_refused_by_the_launcher_itself():
    """`--gpus 8` on a node that shows fewer devices: the launcher refuses before any rank starts (exit code 2, no line)"""
    env = _env()
    env.pop("QNNP_BENCH_STUB")
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this box has eight GPUs: the refusal cannot be provoked")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 2, (res.returncode, res.stderr[-500:])
    assert "--gpus 8 requested" in res.stderr and "refusing" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_secondary_block_is_flat_and_carries_every_config():
    """roofline.secondary: flat scalars only (the driver's record truncates nothing it does not have to walk), one group
    of keys per BASELINE config, absent sources simply absent"""
    sys.path.insert(0, ROOT)
    import bench
    extra = {
        "q8fc_m1_k1024_n1000": {"kernel": "q8_pw_stream_mfma", "us": 4.2, "weight_gbs": 243.0},
        "q8conv_3x3_56x56x64_b128": {"kernel": "q8_conv_wave_ws_c_mfma", "ms": 0.02, "tops": 1480.0, "gbs": 2569.0,
                                     "frac_of_copy_kernel": 0.5, "roofline_ms": 0.0064},
        "q8dwconv_mobilenetv2_layers": {"hbm_gbs": 3824.0, "ms": 0.16, "frac_of_hbm_peak": 0.478, "frac_of_copy_kernel": 0.8},
        "mobilenetv2_sweep": {"images_per_s": 320000.0, "images_per_s_by_sum_of_layers": 336000.0, "frac_of_hbm_peak": 0.409,
                              "batch_per_gpu": 128},
        "mobilenetv2_network_fused": {"images_per_s": 299000.0},
        "q8gemm_4096_variants": {"kernel_zero_point_126": {"frac": 0.46, "us": 59.0}},
        "q8dwconv_5x5_dilated_and_realistic_scale": {"dw5x5_56x56x72_s2": {"frac_of_hbm_peak": 0.3}},
        "next_rows": {"q8deconv_3x3s2_28x28x64_32": {"gbs": 1700.0}},
        "conv_lists": {"resnet18": {"images_per_s_by_sum_of_layers": 50000.0, "frac_of_bound": 0.2, "worst_dense_3x3_frac": 0.1}},
    }
    sec = bench.secondary_block(extra)
    assert all(not isinstance(v, (dict, list)) for v in sec.values())
    assert sec["c0_fc_m1_k1024_n1000_us"] == 4.2
    assert sec["c2_conv3x3_56x56x64_b128_ms"] == 0.02 and sec["c2_frac_of_bound"] == 0.32
    assert sec["c3_frac_of_hbm_peak"] == 0.478
    assert sec["c4_sweep_images_per_s_graph"] == 320000.0 and sec["c4_sweep_images_per_s_sum_of_layers"] == 336000.0
    assert sec["network_fused_images_per_s"] == 299000.0 and "network_images_per_s" not in sec
    assert sec["gemm4096_kzp126_frac"] == 0.46 and sec["dw5x5_s2_frac_of_hbm_peak"] == 0.3
    assert sec["deconv3x3s2_frac_of_hbm_peak"] == round(1700.0 / 8000.0, 4)
    assert sec["resnet18_worst_dense_3x3_frac_of_bound"] == 0.1
    assert bench.secondary_block({}) == {}
    # the shape lists are the reference's (bench/convolution.cc:642-718, 147-184): row counts and the rows the review named
    assert len(bench.RESNET18) == 11 and len(bench.RESNET50) == 23 and len(bench.SHUFFLENET_V1_G2) == 19
    assert (28, 28, 3, 3, 1, 1, 1, 128, 128) in bench.RESNET18 and (7, 7, 3, 3, 1, 1, 1, 512, 512) in bench.RESNET50
    assert (224, 224, 7, 7, 2, 1, 1, 3, 64) == bench.RESNET18[0] == bench.RESNET50[0]
