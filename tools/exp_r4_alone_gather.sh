# GPU box: the gather ALONE (commit waited for first) per rows-per-workgroup and kernel form: what could the LAST chunk's
# gather -- the tail of a step -- run at?
for v in "A=1" "ZIP_HIP_GATHER_RPB=16" "ZIP_HIP_GATHER_RPB=32" "ZIP_HIP_GATHER_RPB=64" "ZIP_HIP_GATHER_RPB=96" "ZIP_HIP_GATHER_RPB=128" "ZIP_HIP_GATHER_STREAM=1" "ZIP_HIP_GATHER_STREAM=1 ZIP_HIP_GATHER_RPB=64" "ZIP_HIP_GATHER_STREAM=1 ZIP_HIP_GATHER_RPB=128" "ZIP_HIP_GATHER_PRIO=0"; do
  echo -n "$v : "; env $v python3 tools/kernel_times.py --hint --serial --reps 6 2>/dev/null | grep open_columns
done
