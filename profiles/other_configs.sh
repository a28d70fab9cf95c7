set -o pipefail
O=gpurun_out/r2l_other.txt
: > $O
python tests/manual/c4_check.py >> $O 2>&1 && python tests/manual/c4_rows_timing.py >> $O 2>&1 && \
python tests/manual/fit_bench.py coat 8 500 400 20 >> $O 2>&1 && \
python tests/manual/fit_bench.py kuairec_small 16 2000 500 5 >> $O 2>&1 && \
python tests/manual/fit_bench.py kuairec_small 400 2000 200 2 >> $O 2>&1 && \
python profiles/eval_bench.py >> $O 2>&1 && \
python profiles/mf_bench.py kuairec_small 16 2000 200 >> $O 2>&1 && \
python profiles/mf_bench.py kuairec_small 100 2000 100 >> $O 2>&1
grep -v "amdgpu.ids" $O | cut -c1-400
