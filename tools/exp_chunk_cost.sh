# GPU box: the commit kernel ALONE (hinted, packed) at 2^24 for 1 .. 16 chunks: what does a chunk end cost?
for n in 1 2 4 0 8 16 1 0; do
  if [ $n = 0 ]; then echo "== default schedule"; python3 tools/kernel_times.py --hint --serial --reps 8 | grep raa_commit
  else echo "== ZIP_HIP_CHUNKS=$n"; ZIP_HIP_CHUNKS=$n python3 tools/kernel_times.py --hint --serial --reps 8 | grep raa_commit; fi
done
