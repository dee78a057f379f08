#!/bin/bash
# debug: ranks as threads of a process over the stub transport (tests/mp_rank.py "r0,r1"); usage: dbg_threads.sh <case> <world> <per_process>
cd $(dirname $0)/..
CASE=${1:-cavity64_b2x2x1}; W=${2:-4}; PP=${3:-2}
[ -f scripts/dbg/libbtdump.so ] || gcc -O1 -g -shared -fPIC -o scripts/dbg/libbtdump.so scripts/dbg/btdump.c -lpthread
OUT=gpurun_out/dbg_threads; rm -rf $OUT; mkdir -p $OUT
[ -n "$DBG_ENV" ] && export $DBG_ENV
export PS_TEST_TRANSPORT=stub PS_RCCL_LIB=$PWD/tests/stub_rccl/libps_stub_rccl.so PS_DIST_OVERLAP=${OVERLAP:-1} PS_FUSED_R=${FUSED:-1} PS_VERBOSE=1 PYTHONFAULTHANDLER=1 PS_DBG_DUMP=25 PS_DBG_BT=$PWD/scripts/dbg/libbtdump.so
pids=""
for ((q=0; q<W; q+=PP)); do
  ranks=""; outs=""
  for ((r=q; r<q+PP; r++)); do ranks="$ranks,$r"; outs="$outs,$OUT/r$r.npz"; done
  timeout -k 5 45 python tests/mp_rank.py $CASE $W ${ranks:1} 31000 ${outs:1} > $OUT/p$q.log 2>&1 &
  pids="$pids $!"
done
for p in $pids; do wait $p; echo "process exit $?"; done
for f in $OUT/p*.log; do echo "== $f"; grep -A14 "^--- thread" $f | cut -c1-160; done
