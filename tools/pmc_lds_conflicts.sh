#!/bin/bash
# GPU box: LDS bank conflicts per kernel over two denoiser evaluations of the benchmark (one --pmc pass, kernel-trace only)
O=/root/repo/gpurun_out/pmc_lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --ddim-steps 2 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-overlap --no-decode --no-max-clip --no-shard-sim --no-other-configs"
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/a -o p -- $B > $O/a.log 2>&1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $O/b -o p -- $B > $O/b.log 2>&1
DBA=$(find $O/a -name "*.db" | head -1); DBB=$(find $O/b -name "*.db" | head -1)
python /root/repo/tools/rocpd_pmc.py $DBA > $O/lds_a.txt 2>&1
python /root/repo/tools/rocpd_pmc.py $DBB > $O/lds_b.txt 2>&1
find $O -name "*.db" -delete
python - <<'PY'
import re, collections
O = "/root/repo/gpurun_out/pmc_lds"
d = collections.defaultdict(dict)
for f in ("lds_a.txt", "lds_b.txt"):
    for line in open(f"{O}/{f}"):
        m = re.match(r"(.{62}) (\S+)\s+n=\s*(\d+) avg=\s*([0-9.]+)", line)
        if m:
            d[m.group(1).strip()][m.group(2)] = (int(m.group(3)), float(m.group(4)))
rows = []
for k, c in d.items():
    if "SQ_LDS_IDX_ACTIVE" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n, act = c["SQ_LDS_IDX_ACTIVE"]; conf = c["SQ_LDS_BANK_CONFLICT"][1]; gui = c["GRBM_GUI_ACTIVE"][1]; wl = c["SQ_WAIT_INST_LDS"][1]; wc = c["SQ_WAVE_CYCLES"][1]
    rows.append((n * gui, k, n, act, conf, gui, wl, wc))
print("| kernel | launches | LDS active / CU, share of the launch | bank-conflict share of LDS active | waves waiting on LDS (share of wave cycles) |")
print("|---|---|---|---|---|")
for _, k, n, act, conf, gui, wl, wc in sorted(rows, reverse=True)[:24]:
    print(f"| {k[:60]} | {n} | {100 * (act / 256) / (gui / 8):.0f} % | {100 * conf / act if act else 0:.0f} % | {100 * wl / wc if wc else 0:.0f} % |")
PY
