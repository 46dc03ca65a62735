#!/bin/bash
# round 6, session 10: the high-entropy variant without a 3-byte chain (LZS_WGV_NO3), by 2-byte bucket count; forced on the other two
# classes as well (any variant must give the same bytes); word-sized dummies for pools of 512 (b1) against the tree before (b0) and round 5's
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT/tools/probes
{
for rep in 1 2; do for v in cur2 n3 n3h2k n3h4k; do echo -n "$v: "; env $( [ $rep = 2 ] && echo AB_NOHASH=1 ) timeout 120 ./ab_$v 2; done; done
for c in 0 1; do for v in cur2 n3h2k; do echo -n "$v [LZS_VARIANT=lit]: "; LZS_VARIANT=lit timeout 300 ./ab_$v $c; done; done
for rep in 1 2; do for v in r5d b0 b1; do echo -n "$v: "; AB_NOHASH=1 timeout 120 ./ab_$v 0; done; done
} > $OUT/ab_s10.txt 2>&1
cat $OUT/ab_s10.txt
