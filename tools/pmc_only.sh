TAG=r2; R=/root/repo; O=$R/gpurun_out
bash $R/tools/pmc_bench.sh $O/${TAG}_pmc > /dev/null 2>&1
F=$(find $O/${TAG}_pmc/fetch -name "*.db" | head -1); W=$(find $O/${TAG}_pmc/write -name "*.db" | head -1)
python $R/tools/pmc_traffic_json.py $F $W $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.log 2>&1
python $R/tools/pmc_hbm_by_kernel.py $F $W 2 > $O/${TAG}_hbm_by_kernel.md 2>&1
find $O/${TAG}_pmc -name "*.db" -delete
head -24 $O/${TAG}_hbm_by_kernel.md
