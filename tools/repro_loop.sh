# loop a pytest selection on the GPU box: bash tools/repro_loop.sh <iterations> <pytest args...>   (output of a failing run is printed; the abort handler of
# tools/diag/abort_backtrace.c is preloaded so that a glibc abort says who called free())
set -u
mkdir -p gpurun_out; cd /root/repo
N=$1; shift
fails=0
for i in $(seq 1 $N); do
  LD_PRELOAD=/root/repo/tools/diag/libabort_backtrace.so timeout 900 python -m pytest "$@" -m gpu -x -q -s -p no:faulthandler > gpurun_out/repro_$i.txt 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then
    fails=$((fails+1))
    echo "iter $i rc $rc"; python -c "import sys; t=open(sys.argv[1],errors='replace').read(); print(t[:200]); print(t[t.find('===='):][:9000])" gpurun_out/repro_$i.txt
    cp gpurun_out/repro_$i.txt gpurun_out/abort_$i.txt
    [ $fails -ge 2 ] && break
  fi
  rm -f gpurun_out/repro_$i.txt
done
echo "failures $fails of $i"
