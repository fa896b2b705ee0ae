#!/bin/bash
# GPU box (round 6): does a small companion upload next to every download keep a download burst from flipping the copy-stream pair into its serialised state?
# Needs the diagnostic builds: tools/build_variants.sh probediag_<mode>_<kb> "-DSE3D_PROBE_DIAG=<mode> -DSE3D_PROBE_COMPANION_KB=<kb>" "" api for the four (mode, KB) pairs below.  -> profiles/r6/stream_companion.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
for cfg in "15 1024" "27 1024" "27 8192" "27 64" "15 1024"; do set -- $cfg
SE3D_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/tmp/libse3d_probediag_$1_$2.so timeout 600 python bench.py --no-secondary --no-serial --no-parity --no-cpu-baseline --no-robust --stream-repeat 4 > /dev/null 2> $O/comp_$1_$2.err
echo "mode $1 companion $2 KB:"; grep "streaming pass" $O/comp_$1_$2.err | sed "s/^streaming pass \([0-9]\): \([0-9.]*\) M.*copy-stream probe {/  pass \1 \2 M /; s/, 'submit_mean.*//"
done
