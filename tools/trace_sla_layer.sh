cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=/root/repo/tools/ubench/libdawn_hip_sla_fp32out.bin; else unset DAWN_HIP_LIB; fi
  rm -rf /tmp/slat_$v; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/slat_$v -o t -- python /root/repo/tools/bench_sla_layer.py > /dev/null 2>&1
  DB=$(find /tmp/slat_$v -name "*.db" | head -1); echo "== $v"; python /root/repo/tools/rocpd_summary.py $DB 2>&1 | grep -E "sla_" | cut -c1-150
done
