#!/bin/bash
# k_ba_schur_pairs_mfma: waves per SIMD the four-wave instantiation is compiled for (EAO_PAIR_OCC), rebuilt ON the GPU box (hipcc is there) and timed with rocprofv3 on the
# 25-window batch in one group, then the default call.   bash tools/sweep_pair_variants.sh "8 6 5"   -> gpurun_out/r04_pair_variants.txt
# (Round 4's other variants of this kernel -- prefetch rings, class-split launches, heaviest-first order, the branch-free loop -- were measured with earlier versions of this
#  script and are not in the tree: profiles/r04_ba_pair_ablation.txt.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_pair_variants.txt
echo "# waves per SIMD -> average launch (us) of the pair kernel, 25 windows, EAO_BA_BATCH_GROUPS=1; and the default call" > $OUT
C=eao_fusion_amd/csrc
for occ in $1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DEAO_PAIR_OCC=$occ -c $C/lm.hip -o $C/build/lm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o eao_fusion_amd/libeaofusion_hip.so $C/build/api_common.o $C/build/orb.o $C/build/hamming.o $C/build/match.o $C/build/search.o $C/build/frame.o $C/build/lm.o $C/build/track.o -ldl -Wl,-rpath,/opt/rocm/lib || exit 1
  rm -rf gpurun_out/pv
  EAO_BA_BATCH_GROUPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pv -o s -- python3 tools/dbg_ba_batch.py > gpurun_out/pv.log 2>&1
  f=$(find gpurun_out/pv -name "*kernel_stats.csv" | head -1)
  line=$(python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pairs_mfma" in r["Name"]]
print(" + ".join("%s %.1f" % (r["Name"].split("pairs_mfma")[1][:4], float(r["AverageNs"]) / 1e3) for r in rows))
PY
)
  call=$(python3 tools/dbg_ba_batch.py 2>&1 | grep "eao_local_ba_batch" | head -1 | sed 's/.*min \([0-9.]*\) median \([0-9.]*\).*/min \1 median \2 ms/')
  echo "$occ waves per SIMD: $line us; default call: $call" | tee -a $OUT
done
# leave the DEFAULT library installed (the loop above overwrote it with the last -DEAO_PAIR_OCC variant)
rm -f $C/build/lm.o && make -C $C
