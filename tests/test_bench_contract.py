"""CPU: the committed bench line (profiles/r03_bench_n1.json, produced by `python bench.py` on an MI355X) carries every
field of the measurement contract; bench.py's command line keeps the documented flags."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(",")[0] == base["metric"].split(",")[0] and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["frames_per_video"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    # round 2: the line names the BASELINE configuration it ran, the layout, and the whole-job fraction of the MFMA peak
    assert "24f@256x256" in d["metric"] and "configs[1]" in d["config"]["workload"] and d["config"]["layout"] == "single"
    wv = r["whole_video"]
    assert abs(wv["frac"] - wv["tflop"] / (d["ms_per_step"] * 1e-3) / d["n_gpus"] / r["peak"]) < 2e-3
    assert r["strict_bytes_per_launch"] < r["algorithmic_bytes_per_launch"] and abs(r["traffic_over_strict"] - r["traffic"] / r["strict_bytes_per_launch"]) < 1e-2
    big = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_n1_125f.json")))
    assert "125f@256x256" in big["metric"] and "configs[2]" in big["config"]["workload"] and big["roofline"]["traffic"] is None
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    # round 3: box calibration either side of the timed region, the real reference's recorded CPU timing, the communicators used
    cal = r["calibration"]
    assert cal["gemm_8192_tflops_before"] > 300 and cal["gemm_8192_tflops_after"] > 300
    ref = c["reference_recorded"]
    assert ref["kind"] == "reference" and ref["value"] > 0 and ref["cores"] >= 1
    assert d["config"]["rccl_communicators"] == []
    lv = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_n1_lvdm.json")))
    assert "VideoCrafter LVDM 16f@256x256" in lv["metric"] and "configs[4]" in lv["config"]["workload"] and lv["value"] > 0
    for name, layout in (("r03_rehearsal_n4_one_gpu.json", "tshard"), ("r03_rehearsal_n2_one_gpu.json", "pairs")):
        rh = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert rh["config"]["layout"] == "replicas" and rh["scaling"] == "weak" and "REHEARSAL" in rh["data"]
        assert rh["collective_layout"]["layout"] == layout and rh["collective_layout"]["value"] > 0


def test_bench_cli_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--parallel", "--videos"):
        assert flag in out.stdout


def test_bench_self_launches_its_ranks_without_an_external_launcher():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own N ranks (the bare form
    used to die on `assert world == args.gpus`).  --launch-check stops after the rendezvous, so this runs without a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 3, "rank_sum": 6}


def test_bench_refuses_more_gpus_than_the_node_has():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "T2V_BENCH_ONE_DEVICE")}
    env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode == 2 and "--gpus 64 but this node exposes" in out.stderr


def test_layout_choice_is_comparable_across_n():
    """The clip is configs[1]'s 24 frames in every layout (BASELINE.json's metric names it), so the per-N values are comparable.
    Round 4: under `auto` the frame-parallel layout of that N is measured first as its own bounded job and is the
    headline; the outer job's own ranks run replicas (side figure / fallback)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.choose_layout(1, "auto") == ("single", 24)
    for n in (2, 3, 4, 8):
        assert bench.choose_layout(n, "auto") == ("replicas", 24)
    assert bench.choose_layout(8, "tshard") == ("tshard", 24) and bench.choose_layout(2, "pairs") == ("pairs", 24)
    assert bench.choose_layout(8, "tshard", 125) == ("tshard", 125)
    assert [bench.collective_layout_of(n) for n in (1, 2, 3, 4, 5, 6, 8)] == [None, "pairs", None, "tshard", None, "tshard", "tshard"]


def _args(**kw):
    import argparse
    base = dict(steps=3, warmup=1, ddim_steps=50, height=256, width=256, frames=0, also_frames=125, collective_timeout=5)
    base.update(kw)
    return argparse.Namespace(**base)


def test_frame_parallel_job_result_and_fallback_reason(monkeypatch, tmp_path):
    """Round 4: the frame-parallel layout is measured by its own bounded job; its JSON line becomes the headline,
    a failure / timeout becomes `layout_fallback.reason` (with the job's last diagnostics), never an exception."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_ok(n, argv, timeout_s=None, capture=False, stderr_path=None):
        seen["argv"], seen["timeout"] = list(argv), timeout_s
        open(stderr_path, "w").write("[Gloo] banner\n")
        return 0, 'noise\n{"value": 55.5, "ms_per_step": 432.1, "config": {"layout": "tshard"}}\n'
    monkeypatch.setattr(bench, "self_launch", fake_ok)
    res = bench.collective_layout_job(8, _args(), 77)
    assert res["ok"] and res["layout"] == "tshard" and res["line"]["value"] == 55.5 and res["timeout_s"] == 77
    a = seen["argv"]
    assert a[a.index("--parallel") + 1] == "tshard" and a[a.index("--steps") + 1] == "3" and a[a.index("--warmup") + 1] == "1"
    assert a[a.index("--also-frames") + 1] == "125" and "--no-collective-job" in a and "--frames" not in a and seen["timeout"] == 77
    assert bench.collective_layout_job(2, _args(warmup=0), 5)["layout"] == "pairs"
    assert seen["argv"][seen["argv"].index("--warmup") + 1] == "1"          # the bounded job always warms up at least once

    def fake_timeout(n, argv, timeout_s=None, capture=False, stderr_path=None):
        open(stderr_path, "w").write("UserWarning: x\n[bench] rank 3: layout 'tshard' failed: RuntimeError: ncclAllGather: unhandled system error\n")
        return 124, ""
    monkeypatch.setattr(bench, "self_launch", fake_timeout)
    res = bench.collective_layout_job(4, _args(), 9)
    assert not res["ok"] and res["exit_code"] == 124 and "timed out after 9s" in res["reason"] and "ncclAllGather" in res["reason"]

    def fake_crash(n, argv, timeout_s=None, capture=False, stderr_path=None):
        open(stderr_path, "w").write("")
        return 1, "{not json"
    monkeypatch.setattr(bench, "self_launch", fake_crash)
    res = bench.collective_layout_job(4, _args(), 9)
    assert not res["ok"] and res["reason"].startswith("exit code 1")


HANDOFF_WORKER = r'''
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import bench
rank = int(sys.argv[2])
import argparse
args = argparse.Namespace(collective_timeout=20)
if rank == 0:
    def job(world, a, timeout):
        time.sleep(1.0)
        return {"ok": True, "layout": "tshard", "line": {"value": 12.5}, "job_s": 1.0, "timeout_s": timeout}
    bench.collective_layout_job = job
print("RES " + json.dumps(bench.collective_first(4, rank, args, time.time())), flush=True)
'''


def test_every_rank_gets_the_frame_parallel_jobs_result():
    """rank 0 runs the bounded job while the other ranks of the launch wait for the hand-off file (same parent, same port)."""
    env = dict(os.environ, MASTER_PORT="29791")
    procs = [subprocess.Popen([sys.executable, "-c", HANDOFF_WORKER, ROOT, str(r)], stdout=subprocess.PIPE, text=True, env=env) for r in (1, 2, 0)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    res = [json.loads(next(ln for ln in o.splitlines() if ln.startswith("RES "))[4:]) for o in outs]
    assert all(r == res[0] for r in res) and res[0]["ok"] and res[0]["line"]["value"] == 12.5
    import tempfile
    os.remove(os.path.join(tempfile.gettempdir(), f"t2v_bench_handoff_{os.getpid()}_29791.json"))


def test_round4_bench_lines_and_rehearsals():
    """Round 4: the committed N = 1 line of the final build, and the one-GPU rehearsals of the N > 1 flow — the frame-parallel layout is
    the headline (strong scaling, self-check recorded, replicas beside it), an injected failure falls back to replicas with the reason."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["layout"] == "single" and "configs[1]" in d["config"]["workload"]
    assert abs(d["value"] - 24 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    assert 13.5 < r["flops_per_unet_step_T"] < 14.2          # algorithmic: the repeated K of the [hi | lo] GEMMs is not counted
    t = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))
    assert t["hbm_bytes_per_launch"] > 5e7 and t["launches"] > 400
    for name, layout in (("n4", "tshard"), ("n2", "pairs"), ("n4_fake_rccl", "tshard")):
        rh = json.load(open(os.path.join(ROOT, "profiles", f"r04_rehearsal_{name}_one_gpu.json")))
        assert rh["config"]["layout"] == layout and rh["scaling"] == "strong" and rh["value"] > 0 and "REHEARSAL" in rh["data"]
        assert rh["steps"] == 3 and rh["warmup"] == 1 and rh["replicas"]["value"] > 0 and rh["collective_job"]["layout"] == layout
        assert "layout_fallback" not in rh["config"]
        if layout == "tshard":
            assert rh["config"]["self_check"]["ok"] and rh["config"]["self_check"]["collective_ops_per_forward"] > 100
    assert json.load(open(os.path.join(ROOT, "profiles", "r04_rehearsal_n4_fake_rccl_one_gpu.json")))["config"]["self_check"]["in_library"] is True
    fb = json.load(open(os.path.join(ROOT, "profiles", "r04_rehearsal_n4_fallback_one_gpu.json")))
    assert fb["config"]["layout"] == "replicas" and fb["scaling"] == "weak" and fb["value"] > 0
    assert fb["config"]["layout_fallback"]["requested_layout"] == "tshard" and "injected failure" in fb["config"]["layout_fallback"]["reason"]
