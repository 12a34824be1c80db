#!/usr/bin/env python3
"""HBM traffic and achieved HBM GB/s per kernel family for the benchmark command, from the two PMC passes of
tools/pmc_bench.sh (FETCH_SIZE pass, WRITE_SIZE pass; gfx950 x2 FETCH correction, see pmc_traffic_json.py) joined with
the kernel durations of the same (traced) runs:
    python tools/pmc_hbm_by_kernel.py <fetch_results.db> <write_results.db> [n_evaluations] > out.md
n_evaluations = UNet evaluations in the profiled command (bench.py --ddim-steps 2 --steps 1 -> 2)."""
import collections, sqlite3, sys


def short(n):
    return str(n).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


def counters(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ci else "name"
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in db.execute("select * from counters_collection"):
        if r[ci["counter_name"]] == counter:
            k = short(r[ci[name_col]])
            per[k][0] += 1
            per[k][1] += float(r[ci["value"]])
    dur = collections.defaultdict(float)
    for n, s, e in db.execute("select name, start, end from kernels"):
        dur[short(n)] += (e - s) * 1e-9
    return per, dur


def main():
    fetch, dur_f = counters(sys.argv[1], "FETCH_SIZE")
    write, dur_w = counters(sys.argv[2], "WRITE_SIZE")
    nev = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    rows = []
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * fetch.get(k, [0, 0.0])[1] * 1024.0            # KiB, gfx950 half-count correction
        wr = write.get(k, [0, 0.0])[1] * 1024.0
        t = 0.5 * (dur_f.get(k, 0.0) + dur_w.get(k, 0.0))
        rows.append((k, max(fetch.get(k, [0])[0], write.get(k, [0])[0]), rd, wr, t))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    tot_b = sum(r[2] + r[3] for r in rows)
    tot_t = sum(r[4] for r in rows)
    print("# HBM traffic per kernel family (PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes; durations from the same traced runs)\n")
    print(f"Profiled command: bench.py --ddim-steps 2 --steps 1 --warmup 0 --no-overlap (256x256, 200 frames): {nev:.0f} UNet "
          f"evaluations + the once-per-clip work.  Total {tot_b / 1e9:.1f} GB = **{tot_b / nev / 1e9:.1f} GB per evaluation** "
          f"({tot_b / nev / 200 / 1e6:.0f} MB per frame-evaluation; SURVEY 8d fused lower bound: 118 MB), "
          f"kernel time {tot_t * 1e3:.1f} ms -> {tot_b / tot_t / 1e12:.2f} TB/s average while kernels run.\n")
    once = sum(r[2] + r[3] for r in rows if r[0].startswith("at::native::"))
    print(f"Of that, {once / 1e9:.1f} GB are the torch kernels of the once-per-clip work (`at::native::*`: frame-invariance check of "
          f"fea, reference-layout conversions, cats), amortised here over {nev:.0f} evaluations instead of 50: the per-evaluation "
          f"kernels alone move **{(tot_b - once) / nev / 1e9:.1f} GB per evaluation = {(tot_b - once) / nev / 200 / 1e6:.0f} MB per "
          f"frame-evaluation**.\n")
    print("| kernel | launches | read GB | written GB | time ms | achieved TB/s | share of bytes |")
    print("|---|---|---|---|---|---|---|")
    for k, n, rd, wr, t in rows:
        if rd + wr < 0.002 * tot_b:
            continue
        print(f"| {k} | {n} | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {t * 1e3:.2f} | {(rd + wr) / t / 1e12 if t else 0:.2f} | {(rd + wr) / tot_b * 100:.1f} % |")


if __name__ == "__main__":
    main()
