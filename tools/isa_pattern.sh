#!/bin/bash
# Print the run-length pattern (MFMA / valu / ds / vmem) of a kernel's instruction stream from a --save-temps .s file:
#   tools/isa_pattern.sh file.s <mangled-name-substring>
f=$1; k=$2
name=$(grep -E "^_Z[A-Za-z0-9_]*${k}[A-Za-z0-9_]*:" $f | head -1 | cut -d: -f1)
echo "kernel $name"
awk -v n="$name:" 'index($0,n)==1{p=1} p{print} p&&/s_endpgm/{exit}' $f | awk '{print $1}' | sed 's/v_mfma.*/MFMA/; s/^v_.*/valu/; s/^s_barrier/BARRIER/; s/^s_.*/salu/; s/^ds_.*/ds/; s/^buffer.*/vmem/; s/^global.*/vmem/' | grep -v "^;" | grep -v "^\." | grep -v salu | uniq -c | awk '{printf "%s%s ", $1, $2} /BARRIER/{print ""}'
echo
