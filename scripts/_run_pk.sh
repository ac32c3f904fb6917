set -o pipefail
timeout -k 10 300 python3 -m pytest tests/test_gpu_grid.py -x -q -k "power_spectrum or config5_size or slab" 2>&1 | tail -3 || exit 1
for T in 512 1024; do for LT in 4 8; do
echo "== threads $T tile $LT"; BFGX_FFT_THREADS=$T BFGX_FFT_TILE=$LT scripts/pk_variants.sh base | grep -v "fwd_len\|unitstride\|table_build\|table_sums\|== base"
done; done
