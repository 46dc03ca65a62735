#!/bin/bash
# round 6, session 11: the 2-byte chain alone (LZS_WGV_NO3) for the high-entropy variant by hops, for the low-entropy variant by buckets
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT/tools/probes
{
for rep in 1 2; do for v in cur2 n3h2k n3h2kh2 n3h2kh4 n3h2kh5; do echo -n "$v: "; env $( [ $rep = 2 ] && echo AB_NOHASH=1 ) timeout 120 ./ab_$v 2; done; done
for rep in 1 2; do for v in cur2 f3 f3h1k f3h2k; do echo -n "$v: "; env $( [ $rep = 2 ] && echo AB_NOHASH=1 ) timeout 120 ./ab_$v 1; done; done
for rep in 1 2; do for v in b0 b2; do echo -n "$v: "; AB_NOHASH=1 timeout 120 ./ab_$v 0; done; done
} > $OUT/ab_s11.txt 2>&1
cat $OUT/ab_s11.txt
