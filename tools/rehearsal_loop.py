"""Reproduce the two-rank one-GPU rehearsal hang (`tests/test_train_gpu.py::test_bench_self_launches_ranks` hung about
once in a dozen suite runs and was papered over with a retry). Loops `bench.py --gpus 2` (tiny dims, gloo collectives on CUDA tensors of
two processes sharing cuda:0 — the rehearsal, not the RCCL product path) N times with TORCH_DISTRIBUTED_DEBUG=DETAIL and short stall
limits: a rank that sits in one stage longer than --stage_timeout dumps the Python stack of every thread (bench.Progress / faulthandler),
the self-launching parent prints every rank's last stage and kills exactly the process group it started. Every failed run's stderr is kept.

    python tools/rehearsal_loop.py --runs 50 --out gpurun_out/rehearsal
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=50)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rehearsal"))
    ap.add_argument("--stage_timeout", type=int, default=40)
    ap.add_argument("--launch_timeout", type=int, default=150)
    ap.add_argument("--budget_s", type=int, default=900, help="stop starting new runs after this many seconds")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    env = dict(os.environ, GROVE_BENCH_BACKEND="gloo", GROVE_BENCH_ONE_GPU="1", GLOO_SOCKET_IFNAME="lo", TORCH_DISTRIBUTED_DEBUG="DETAIL",
               TORCH_CPP_LOG_LEVEL="INFO")
    env.pop("WORLD_SIZE", None)
    arms = [[], ["--no_comm_overlap"], ["--exchange", "rs_ag"], ["--exchange", "a2a_f32"], ["--dense_embed"],
            ["--mode", "infer", "--frames", "16", "--batch", "1"]]
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dims", "tiny", "--steps", "2", "--warmup", "1", "--frames", "8",
            "--text_len", "48", "--no_cpu_baseline", "--stage_timeout", str(args.stage_timeout), "--launch_timeout", str(args.launch_timeout)]
    t_start = time.time()
    rec = []
    for i in range(args.runs):
        if time.time() - t_start > args.budget_s:
            break
        extra = arms[i % len(arms)]
        t0 = time.time()
        p = subprocess.run(base + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        dt = time.time() - t0
        lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
        ok = p.returncode == 0 and len(lines) == 1
        rec.append({"run": i, "arm": " ".join(extra) or "default", "rc": p.returncode, "seconds": round(dt, 1), "ok": ok})
        print(rec[-1], flush=True)
        if not ok:
            with open(os.path.join(args.out, f"fail_{i:03d}.stderr.txt"), "w") as fh:
                fh.write(p.stderr[-200000:])
            with open(os.path.join(args.out, f"fail_{i:03d}.stdout.txt"), "w") as fh:
                fh.write(p.stdout[-20000:])
    summary = {"runs": len(rec), "ok": sum(r["ok"] for r in rec), "failed": [r for r in rec if not r["ok"]],
               "seconds_mean": round(sum(r["seconds"] for r in rec) / max(len(rec), 1), 1), "seconds_max": max((r["seconds"] for r in rec), default=0),
               "records": rec}
    with open(os.path.join(args.out, "summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "records"}))


if __name__ == "__main__":
    main()
