"""the scaling curve that cannot be measured (one GPU per lease), PREDICTED from runs whose N ranks share device 0.

    python scripts/mg_predict.py [c3 c4e] > gpurun_out/round6_mg_predicted.md

Per workload and N in 1, 2, 4, 8: bench.py --gpus N with BFC_BENCH_DEVICES=0,0,... (all ranks in this process, records by the push kernel between ranks
that share the device), strong scaling, verified against the reference's answers.  Reported: the step's wall time on the one device -- all N ranks' kernels side by side plus the copies that stand in for the links: what N real
GPUs would each spend at most 1/N of (a rank's own HIP-event stage times are useless here: they contain the neighbours' kernels) -- against the 1-GPU
run's: the work inflation of owner-computes (N times smaller batches per rank, every rank sweeping its share of the filter per global batch), the bytes a rank puts on its links per step (as sent, and the live records alone), the time those take on N - 1 xGMI links
at 153 GB/s and at half of it, and the step time and efficiency that follow if the exchange overlaps the kernels as designed.  A model, not a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEAK = os.environ.get("MG_SCALING") == "weak"  # one read set PER RANK into one filter of 2^(b + log2 N) bits (capped at -b37): the work per GPU is fixed, a step should take what one GPU's takes
wls = sys.argv[1:] or ["c3", "c4e"]
EXTRA = {"c4e": ["--batch-reads", "8388608"]} if not WEAK else {"c3": ["--batch-reads", "2097152"]}  # (weak: N whole read sets and N full-size contexts on ONE device -- calls of 2.1 M reads per rank, the N = 1 row too)  # (N contexts on ONE device: calls of 8.4 M reads instead of 16.8 M so that their buffers fit its 288 GB; the N = 1 row runs the same calls)
out = []
for wl in wls:
    rows = []
    base = None
    for n in (1, 2, 4, 8):
        env = dict(os.environ)
        if n > 1:
            env["BFC_BENCH_DEVICES"] = ",".join(["0"] * n)
        for tr in ((0,) if n == 1 else (0, 2)):  # default (push kernel: exact bytes) and whole-block peer copies
            env["BFC_BENCH_TRANSPORT"] = str(tr)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", wl, "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                                "--no-boundary", "--no-secondary"] + (["--scaling", "weak"] if WEAK and n > 1 else []) + EXTRA.get(wl, []), capture_output=True, text=True, env=env)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:  # noqa: BLE001
                print("run failed:", wl, n, tr, r.stderr[-400:], file=sys.stderr)
                continue
            if d.get("error"):
                print("run failed:", wl, n, tr, d["error"], file=sys.stderr)
                continue
            st = d["config"]["stage_ms_per_step"]
            m = d.get("multi_gpu_model") or {}
            ksum = d["ms_per_step"]  # what ONE device needs for all N ranks' work (their kernels side by side + the copies that emulate the links): wall time
            if n == 1:
                base = ksum
            rows.append(dict(n=n, transport=m.get("transport", "-"), ms=d["ms_per_step"], verified=d.get("verified"), ksum=ksum, stage={k: round(v * (n if n > 1 else 1), 1) for k, v in st.items()},
                             links=(m.get("exchange_bytes_per_rank_per_step") or {}).get("links", 0), exact=(m.get("exchange_bytes_per_rank_per_step") or {}).get("exact", 0),
                             x_ms=m.get("xgmi_ms_per_step_at_link_peak") or {}, batches=d["config"]["batches_per_step"], lib=d["config"]["library_batches_per_step"]))
    out.append((wl, base, rows))
    print("[mg_predict]", wl, "done", file=sys.stderr, flush=True)

print("# Round 6 -- the scaling curve PREDICTED from ranks emulated on one MI355X (scripts/mg_predict.py; a model, not a measurement)\n")
for wl, base, rows in out:
    print(("## %s, WEAK scaling (one read set per rank -- other genomes: seed + rank -- into ONE filter of 2^(b + log2 N) bits, at most -b37; no reference answers exist for the union: not verified; work inflation = wall over N x the one-GPU wall)\n" if WEAK else
           "## %s, strong scaling (one read set, every global batch split over the ranks), every run verified against the reference's answers\n") % wl)
    print("| N | transport | verified | wall ms per step, all N ranks on ONE device | work inflation vs N = 1 | device ms per rank on N GPUs (wall / N) | bytes out per rank and step: as sent / live records | "
          "their time on N - 1 links at 153 GB/s: as sent / live | predicted ms per step (exchange at half the link rate) | predicted speed-up over one GPU | efficiency |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        n = r["n"]
        infl = r["ksum"] / base if base else float("nan")
        kr = r["ksum"] / n
        xl, xe = r["x_ms"].get("links", 0.0), r["x_ms"].get("exact", 0.0)
        pred = max(kr, 2 * xl)
        sp = (n * base / pred if WEAK else base / pred) if base else float("nan")  # (weak: N GPUs do N times the work in `pred`)
        if WEAK:
            infl = r["ksum"] / (n * base) if base else float("nan")
        print("| %d | %s | %s | %.1f | %.2f | %.1f | %.2f / %.2f GB | %.1f / %.1f ms | %.1f | %.2f | %.2f |" % (
            n, r["transport"], r["verified"], r["ksum"], infl, kr, r["links"] / 1e9, r["exact"] / 1e9, xl, xe, pred, sp, sp / n))
    print()
    for r in rows:
        print("* N = %d (%s): stage ms summed over the ranks %s; %d global batches, %.1f library batches per rank and step" % (r["n"], r["transport"], r["stage"], r["batches"], r["lib"]))
    print()
