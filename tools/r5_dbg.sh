cd $GRAFT_REPO_ROOT
timeout -k 10 300 /opt/rocm/bin/rocgdb -batch -ex run -ex "info registers rdi rsi rdx rcx rax" -ex "x/3i $pc" -ex "info proc mappings" --args python -m pytest tests/test_align_gpu.py -x -q -m gpu -k "test_single_pair_bit_exact" > gpurun_out/r5r_gdb.log 2>&1
grep -n "SIGSEGV" -A 12 gpurun_out/r5r_gdb.log | head -30; grep -n "hipHostMalloc\|kfd\|\[heap\]" gpurun_out/r5r_gdb.log | head
