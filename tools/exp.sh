timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q -k "large_codeword or 2pow26 or for_open or commit_bit_exact" 2>&1 | tail -3
ZIP_HIP_OLD16=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q -k "large_codeword or 2pow26" 2>&1 | tail -2
for i in 1 2; do
echo "== new"; timeout -k 10 300 python tools/kernel_times.py --num-vars 26 --reps 3 --serial 2>&1 | grep -E "raa_commit|raa_encode"
echo "== old"; ZIP_HIP_OLD16=1 timeout -k 10 300 python tools/kernel_times.py --num-vars 26 --reps 3 --serial 2>&1 | grep -E "raa_commit|raa_encode"
done
