for z in 0 2; do
if [ $z = 0 ]; then unset ZERO_INPUTS; else export ZERO_INPUTS=$z; fi
for lib in "" $(ls tools/probe/lib_*.so); do
  echo "== zero=$z lib=$lib"
  if [ -n "$lib" ]; then export MRN_LIB_PATH=$PWD/$lib; else unset MRN_LIB_PATH; fi
  NSHAPES=1 python tools/bench_wino.py 10 6 --only-wino 2>/dev/null
done; done
unset MRN_LIB_PATH ZERO_INPUTS
MRN_WINO_ROWS=0 NSHAPES=1 python tools/bench_wino.py 10 2,6 --only-wino 2>/dev/null
NSHAPES=1 python tools/bench_wino.py 10 2 --only-wino 2>/dev/null
