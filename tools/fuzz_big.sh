# exploration run of the fuzz tests over a seed range (default 1200:4200), audit of the skipped draws: tools/fuzz_big.sh [a:b]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
O=gpurun_out/r06
R=${1:-1200:4200}
rm -f $O/fuzz_skips2.jsonl
(time ODINN_FUZZ_AUDIT=$PWD/$O/fuzz_skips2.jsonl ODINN_FUZZ_SEEDS=$R timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 --timeout 300 -p no:cacheprovider) > $O/fuzz_big_pytest.txt 2>&1
tail -15 $O/fuzz_big_pytest.txt
python tools/fuzz_audit.py $O/fuzz_skips2.jsonl > $O/fuzz_skips2.txt 2>&1; head -12 $O/fuzz_skips2.txt
