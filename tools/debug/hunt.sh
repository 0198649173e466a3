#!/bin/bash
# Diagnostic ("a nondeterminism met on the way"): is this box one where the control build
# fails?  then the candidate libraries, several times each.  The control is csrc/libbxh_ab.so: a development
# build (-DBXH_DEV_ONLY_NP=5 -DBXH_DEV_KEEP_WIDE) of the sources in which MinStd::shuffle is a CALL out of the
# wide kernel's one-lane blocks (bxh_wide.hpp: shuffle_inl -> shuffle); it is not kept in the tree.
# Usage (on the GPU box): bash tools/debug/hunt.sh libbxh.so [more libraries in baryonyx_amd/csrc]
L=$PWD/baryonyx_amd/csrc
run() { BXH_LIB=$1 timeout -k 10 100 python tools/debug/wide_case.py 2>&1 | grep -c "x words differ 0"; }
echo "control: $(BXH_WIDE_TIEDFAST=0 run $L/libbxh_ab.so) $(BXH_WIDE_TIEDFAST=0 run $L/libbxh_ab.so) of 6"
for v in "$@"; do
  echo "$v: $(run $L/$v) $(run $L/$v) $(run $L/$v) $(BXH_WIDE_TIEDFAST=0 run $L/$v) $(BXH_WIDE_TIEDFAST=0 run $L/$v) of 6"
done
