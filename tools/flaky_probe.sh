# repeat the GPU suite under one schedule override and list only failures (a rare failure seen once in tools/suite_matrix.sh)
S=${1:-vjph_strip=0,vjpth_strip=0,dhdt_strip=0}; N=${2:-8}; shift 2
for i in $(seq 1 $N); do
ODINN_SCHEDULE=$S python -m pytest ${@:-tests} -m gpu -q -n 6 --deselect tests/test_gpu_schedule.py --deselect tests/test_gpu_determinism.py -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|non-finite|Error" | cut -c1-300 | head -8
done
