# the full GPU suite up to test_gpu_error_isolation's first test under rocgdb, until glibc aborts: backtraces of every thread at the abort
set -u
mkdir -p gpurun_out; cd /root/repo
export MOT_TEST_STOP_AFTER='test_f3_checksums_do_not_depend_on_the_thread_count[deepocsort]'
for i in $(seq 1 ${1:-12}); do
  timeout 900 rocgdb -q -batch -ex "set pagination off" -ex "set disable-randomization off" -ex "handle SIGABRT stop print" -ex run -ex "bt 40" -ex "info registers rdi rsi" -ex "thread apply all bt 16" --args python -m pytest tests -m gpu -x -q -s > gpurun_out/gdb_$i.txt 2>&1
  if grep -aq "SIGABRT\|double free\|corruption" gpurun_out/gdb_$i.txt; then
    echo "iter $i ABORT"; grep -av "^\[New Thread\|^\[Thread\|^\[Detaching\|^warning" gpurun_out/gdb_$i.txt | head -c 14000; cp gpurun_out/gdb_$i.txt gpurun_out/gdb_abort.txt; break
  else
    echo "iter $i ok: $(grep -a 'passed\|Exit' gpurun_out/gdb_$i.txt | tail -1 | cut -c1-120)"; rm -f gpurun_out/gdb_$i.txt
  fi
done
