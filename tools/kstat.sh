#!/bin/bash
# resource usage + instruction mix of one kernel of a HIP object: tools/kstat.sh <obj.o> <kernel-name-substring>
set -e
T=$(mktemp -d)
python "$(dirname "$0")/extract_co.py" "$1" $T/k.co > /dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co > $T/notes.txt
python - "$T/notes.txt" "$2" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S*%s\S*)' % re.escape(sys.argv[2]), t):
    a = t.rfind('- .agpr_count', 0, m.start()); b = t.find('- .agpr_count', m.end())
    blk = t[a:b if b > 0 else len(t)]
    vals = []
    for k in ['.sgpr_count', '.vgpr_count', '.sgpr_spill_count', '.vgpr_spill_count', '.private_segment_fixed_size', '.group_segment_fixed_size']:
        mm = re.search(re.escape(k) + r':\s+(\d+)', blk); vals.append('%s=%s' % (k[1:], mm.group(1) if mm else '?'))
    print(m.group(1)[:90], ' '.join(vals))
PY
SYM=$(/opt/rocm/lib/llvm/bin/llvm-nm $T/k.co | grep " T " | grep "$2" | head -1 | awk '{print $3}')
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$SYM $T/k.co > ${3:-$T/k.s}
S=${3:-$T/k.s}
echo "$SYM: lines $(wc -l < $S) VALU $(grep -c '\sv_' $S) SALU $(grep -c '\ss_' $S) LDS $(grep -c '\sds_' $S) VMEM $(grep -c 'global_\|flat_\|buffer_' $S) scratch $(grep -c 'scratch_' $S)"
