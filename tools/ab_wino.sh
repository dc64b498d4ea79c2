run() { name=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-isolated-pass 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(r['value'],1), round(r['ms_per_step'],2))"; }
run default X=1
run mincin128 MRN_WINO_MIN_CIN=128
run halves2 MRN_EXPERT_HALVES=2
run halves1 MRN_EXPERT_HALVES=0
run stages4 MRN_LIB_PATH=$PWD/tools/probe/libmrn_MRN_WINO_STAGES_4.so
run nowino MRN_WINO=0
run default_again X=1
