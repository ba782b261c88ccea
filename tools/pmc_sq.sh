#!/bin/bash
# SQ issue/wait breakdown per kernel dispatch (one --pmc pass): tools/pmc_sq.sh <name-filter> -- <command...>
# WAIT_ANY = parked on s_waitcnt/barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing (fractions of wave cycles)
FILT="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp
O=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/pmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $O -o n -- "$@" > $O/log.txt 2>&1
F=$(find $O -name "*counter_collection.csv" | head -1)
python - "$F" "$FILT" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
filt=sys.argv[2]
agg=collections.OrderedDict()
for r in rows:
    k=(int(r['Dispatch_Id']), r['Kernel_Name'])
    d=agg.setdefault(k,{})
    d[r['Counter_Name']]=float(r['Counter_Value'])
    d['_t']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for k,v in agg.items():
    if filt not in k[1]: continue
    wc=max(1.0,v.get('SQ_WAVE_CYCLES',1))
    name=k[1].split('(')[0][-46:]
    print(f"{k[0]:4d} {name:46s} {v['_t']:8.3f} ms  wait={v.get('SQ_WAIT_ANY',0)/wc:.2f} stall={v.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} ldsstall={v.get('SQ_WAIT_INST_LDS',0)/wc:.3f} bankconf={v.get('SQ_LDS_BANK_CONFLICT',0)/max(1,v.get('SQ_LDS_IDX_ACTIVE',1)):.2f} valu={v.get('SQ_INSTS_VALU',0):.3g} wavecyc={wc:.3g}")
PY
rm -rf $O
