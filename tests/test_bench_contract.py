"""bench.py keeps the driver's contract: ONE JSON line from rank 0 with the agreed keys (small config, GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "sfno_debug", "--steps", "2", "--warmup", "1",
                          "--no-sht-metric"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6          # B = 1: samples/s = 1 / step time
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


@pytest.mark.gpu
def test_bench_multistep_rollout_runs():
    """--multistep-count 3 --multistep-checkpoint (makani's multistep flags, BASELINE configs[4] shape of work)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "sfno_debug", "--steps", "2", "--warmup", "1",
                          "--no-sht-metric", "--no-cpu-baseline", "--multistep-count", "3", "--multistep-checkpoint"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["multistep_count"] == 3 and d["config"]["multistep_checkpoint"] is True
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_headline_split_and_secondary_dp():
    """``python bench.py --gpus 2 --parallelism h2w1`` (no torchrun): the script spawns its two ranks, measures the spatial split
    (h2w1, strong scaling) and then data parallelism as ``secondary`` (the DEFAULT at 2 GPUs is data parallelism alone:
    bench.default_parallelism).  Functional run on ONE GPU: both ranks share cuda:0 and exchange through gloo (host-staged)."""
    env = dict(os.environ, MAKANI_AMD_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "sfno_debug", "--steps", "2",
                          "--warmup", "1", "--parallelism", "h2w1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "dp1_h2w1", d["config"]
    assert d["config"]["global_batch"] == 1 and d["value"] > 0 and "note" not in d
    s = d["secondary"]
    assert s["parallelism"] == "dp2" and s["scaling"] == "weak" and s["global_batch"] == 2 and s["value"] > 0
    assert d["cpu_baseline"] is None


@pytest.mark.gpu
def test_bench_eight_ranks_h4w2_line_shape():
    """``python bench.py --gpus 8`` as the driver's SCALE run will issue it, functionally on ONE GPU (8 ranks share cuda:0,
    gloo): the headline is the north-star split h4 x w2 over all 8 ranks running the FUSED exchange schedule (the grid's row
    lengths have segmented FFT kernels), the line names the collectives backend, the exchange volume per step and rank, and
    carries data parallelism over 8 ranks as ``secondary``"""
    env = dict(os.environ, MAKANI_AMD_BENCH_BACKEND="gloo", MAKANI_AMD_BENCH_PHASE_TIMEOUT="800")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "sfno_debug_seg", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=1700, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["parallelism"] == "dp1_h4w2", d["config"]
    assert d["config"]["collectives"] == "gloo" and d["config"]["global_batch"] == 1 and d["value"] > 0 and "note" not in d, d.get("note")
    ex = d["exchange_per_step_rank0"]
    assert ex["schedule"].startswith("fused") and ex["total_MB_sent"] > 0
    assert ex["spatial_group_of_8"]["all_to_alls"] > 0            # the single h x w exchange between FFT and Legendre transform
    assert ex["polar_group_of_4"]["all_to_alls"] > 0 and ex["azimuth_group_of_2"]["all_to_alls"] > 0
    s = d["secondary"]
    assert s["parallelism"] == "dp8" and s["scaling"] == "weak" and s["global_batch"] == 8 and s["value"] > 0
    assert d["cpu_baseline"] is None


@pytest.mark.gpu
def test_bench_rank_launched_by_torchrun_form():
    """the driver's form: ``python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`` — every rank's copy
    of the script starts its own worker per phase on a fresh rendezvous port; rank 0 prints the line"""
    env = dict(os.environ, MAKANI_AMD_BENCH_BACKEND="gloo")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config",
                          "sfno_debug", "--steps", "2", "--warmup", "1", "--parallelism", "h1w2", "--no-secondary"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp1_h1w2" and "secondary" not in d


def _rec(launches, ms_avg, flops_per, bytes_per):
    return dict(launches=launches, ms_avg=ms_avg, ms_total=launches * ms_avg, flops=launches * flops_per, bytes=launches * bytes_per)


def test_roofline_report_on_a_synthetic_profile():
    """the pure reporting functions of bench.py: the dominant kernel is chosen per kernel FAMILY (the nine shapes of the
    channel-GEMM weight gradient are one kernel symbol in the rocprofv3 table), priced against the roof it sits under"""
    sys.path.insert(0, ROOT)
    import bench
    prof = {
        "conv1x1_wgrad_m384_k768_n115200": _rec(70, 0.129, 2.0 * 384 * 768 * 115200, 2.0 * 115200 * (384 + 768)),
        "conv1x1_wgrad_m384_k384_n1038240": _rec(30, 0.539, 2.0 * 384 * 384 * 1038240, 2.0 * 1038240 * (384 + 384)),
        "dhconv_dgrad": _rec(80, 0.330, 68.23e9, 638.5e6),
        "rfft_1440": _rec(30, 0.55, 0.0, 1331e6),
    }
    assert bench.kernel_family("conv1x1_wgrad_m384_k768_n115200") == "conv1x1_wgrad"
    assert bench.kernel_family("legendre_analysis_k240") == "legendre_analysis_k240"
    fam, members = bench.dominant_family(prof)
    assert fam == "dhconv_dgrad" and members == ["dhconv_dgrad"]          # 26.4 ms vs 9.0 + 16.2 = 25.2 ms
    prof["conv1x1_wgrad_m768_k384_n115200"] = _rec(70, 0.127, 2.0 * 384 * 768 * 115200, 2.0 * 115200 * (384 + 768))
    fam, members = bench.dominant_family(prof)
    assert fam == "conv1x1_wgrad" and len(members) == 3

    r = bench.roofline_of(fam, members, prof, "x6", None)
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes"):
        assert k in r, k
    assert r["kernel"] == "conv1x1_wgrad" and r["shapes"] == members and r["traffic"] is None
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0          # 250-330 flop/B: under the ridge
    tot_b = sum(prof[m]["bytes"] for m in members)
    tot_ms = sum(prof[m]["ms_total"] for m in members)
    assert abs(r["achieved"] - tot_b / tot_ms / 1e6) < 0.1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3

    d = bench.roofline_of("dhconv_dgrad", ["dhconv_dgrad"], prof, "x6", 473456640)
    assert d["bound"] == "mfma" and abs(d["peak"] - 2500.0 / 6) < 0.1 and d["traffic"] == 473456640
    assert abs(d["achieved"] - 68.23e9 / 0.330e-3 / 1e12) < 0.01 and "shapes" not in d
    assert abs(bench.roofline_of("dhconv_dgrad", ["dhconv_dgrad"], prof, "fp32", None)["peak"] - 157.3) < 1e-6
    f = bench.roofline_of("rfft_1440", ["rfft_1440"], prof, "x6", 1527091200)
    assert f["bound"] == "hbm" and abs(f["achieved"] - 1331e6 / 0.55e-3 / 1e9) < 0.1
    assert bench.roofline_of("nothing", ["nothing"], prof, "x6", None) is None

    t = bench.kernel_table({}, prof, 10)
    assert t["dhconv_dgrad"]["launches_per_step"] == 8 and abs(t["dhconv_dgrad"]["ms_per_step"] - 2.64) < 1e-9
    warm = {"adamw": _rec(8, 0.39, 0.0, 1.98e9), **{k: _rec(v["launches"] // 10, v["ms_avg"], v["flops"] / v["launches"], v["bytes"] / v["launches"]) for k, v in prof.items()}}
    t = bench.kernel_table(warm, {"dhconv_dgrad": prof["dhconv_dgrad"]}, 10)
    assert t["adamw"]["launches_per_step"] == 8 and t["dhconv_dgrad"]["launches_per_step"] == 8
    # HBM traffic comes from ONE committed file, this round's; a missing file reports nothing instead of an older round's numbers
    pm = bench.load_pmc_traffic()
    assert isinstance(pm, dict) and all(isinstance(v, (int, float)) for v in pm.values())
    cur = bench.PMC_TRAFFIC
    try:
        bench.PMC_TRAFFIC = "r03_pmc_hbm_traffic.json"
        assert bench.load_pmc_traffic().get("dhconv_dgrad", 0) > 0
        bench.PMC_TRAFFIC = "no_such_round.json"
        assert bench.load_pmc_traffic() == {}
    finally:
        bench.PMC_TRAFFIC = cur


@pytest.mark.gpu
def test_bench_fcn3_workload_runs_the_ensemble_recipe():
    """--config fcn3_debug: the FourCastNet3 workload of BASELINE configs[3] on a small grid — ensemble members in the model's
    batch, fair CRPS + 0.1 x spectral CRPS, clip + FusedAdamW; the roofline object names a kernel of the step"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "fcn3_debug", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("FourCastNet3 train samples/sec") and d["unit"] == "samples/s" and d["value"] > 0
    assert d["config"]["workload"] == "fcn3_debug" and d["config"]["ensemble_size"] == 2 and d["config"]["global_batch"] == 1
    assert d["final_loss"] == d["final_loss"] and d["roofline"]["bound"] in ("hbm", "mfma", "valu")
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    # the line's cpu_baseline (the oracle's processor blocks on the host) and its in-run parity: the measured model's first global
    # and first local block against the oracle passes that were timed
    c, par = d["cpu_baseline"], d["parity_rel_l2"]
    print("fcn3_debug cpu_baseline:", c["value"], c["measured"], "| parity:", {k: v for k, v in par.items() if k != "what"})
    assert c["kind"] == "port" and c["value"] > 0 and c["measured"].startswith("processor blocks")
    assert par["global_block_fp32"] < 1e-4 and par["local_block_fp32"] < 1e-4
    assert par["global_block_bf16_autocast"] < 2e-2 and par["local_block_bf16_autocast"] < 2e-2


def test_roofline_prices_the_disco_contraction_against_the_vector_peak():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.kernel_family("disco_fwd_360x720_p1354") == "disco_fwd"
    prof = {"disco_fwd_360x720_p1354": _rec(8, 7.0, 2.0 * 1354 * 720 * 585036, 2.0 * 1354 * 259200 * 10)}
    r = bench.roofline_of("disco_fwd", ["disco_fwd_360x720_p1354"], prof, "x6", None)
    assert r["bound"] == "valu" and r["peak"] == 157.3 and abs(r["achieved"] - 2.0 * 1354 * 720 * 585036 / 7.0e-3 / 1e12) < 0.01
    assert bench.default_parallelism(8, "fcn3_sc2_edim45_layers10") == "h2w2" and bench.default_parallelism(8) == "h4w2"


def test_fcn3_cpu_baseline_child_on_the_small_stand_in():
    """the FourCastNet3 line's ``cpu_baseline``: the child process times the oracle's global and local processor blocks (forward,
    forward + backward) and the parent turns the records into the baseline object — run here on the small stand-in configuration
    (the benchmark calls it for configs[3] only: there the blocks are 360 x 720 x 677 channels)"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.cpu_baseline_fcn3("fcn3_debug", timeout_s=300)
    assert r["kind"] == "port" and r["unit"] == "samples/s" and r["value"] > 0 and r["cores"] >= 1
    assert r["measured"] == "processor blocks, fwd + bwd" and "NOT included" in r["sample"] and "32x64" in r["sample"]
    assert abs(r["ms_per_step"] * 1e-3 * r["value"] - 1.0) < 1e-9
    assert bench.cpu_baseline("sfno_debug") is None


def test_fcn3_cpu_child_is_the_oracle_side_of_the_block_parity(tmp_path):
    """with the weights of the measured model's blocks handed over (bench.Fcn3ParityProbe writes them from the GPU model; here the
    oracle's own blocks stand in), the child's timed forward passes run THOSE weights on the seeded parity input and their outputs
    come back: bit-identical to running the same blocks here"""
    import types
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import fcn3 as ofc, sht as osht
    m = bench.CONFIGS["fcn3_debug"]["model"]
    h, w = m["inp_shape"][0] // m["scale_factor"], m["inp_shape"][1] // m["scale_factor"]
    _, _, _, _, levels = ofc.get_channel_groups(m["channel_names"], m["aux_channel_names"])
    total = len(levels) * m["atmo_embed_dim"] + m["surf_embed_dim"]
    cin = total + m["aux_embed_dim"]
    sht = osht.RealSHT(h, w, lmax=h, mmax=w // 2 + 1, grid="legendre-gauss").float()
    isht = osht.InverseRealSHT(h, w, lmax=h, mmax=w // 2 + 1, grid="legendre-gauss").float()
    torch.manual_seed(5)
    blocks = {k: ofc.NeuralOperatorBlock(sht, isht, cin, total, conv_type=k, mlp_ratio=m["mlp_ratio"], normalization_layer="none", use_mlp=True,
                                         kernel_shape=(3, 3), basis_type="morlet") for k in ("global", "local")}
    with torch.no_grad():
        for b in blocks.values():
            b.layer_scale.weight.normal_()
    parity = types.SimpleNamespace(state_path=str(tmp_path / "blocks.pt"), out_path=str(tmp_path / "out.pt"), oracle_bf16=None)
    torch.save({k: b.state_dict() for k, b in blocks.items()}, parity.state_path)
    r = bench.cpu_baseline("fcn3_debug", timeout_s=300, parity=parity)
    assert r["value"] > 0 and r["measured"] == "processor blocks, fwd + bwd"
    got = torch.load(parity.out_path)
    x = bench.fcn3_parity_input(cin, h, w)
    with torch.no_grad():
        for k, b in blocks.items():
            assert torch.equal(got[k], b(x)), k
