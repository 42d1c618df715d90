# Host side of the product library under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: the pool has no GPU ASan; the device
# code is compiled as usual and not run).  Exercises every entry the CPU tests reach: host field / curve / pairing code, verifiers,
# transcripts, the FRI host path, argument checking.  ~4 min to build.   bash tools/host_sanitizers.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/libzkp_hip_asan.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -Wno-unused-result -pthread \
    -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer "$R/zkp-implementation_amd/csrc/api.hip" -o "$OUT"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$R"
ZKP_HIP_LIB=$OUT LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 \
    python -m pytest tests/test_pairing_cpu.py tests/test_fri_host_cpu.py tests/test_abi_cpu.py tests/test_host_ff_cpu.py tests/test_plonk_model.py -x -q -s \
    > "${TMPDIR:-/tmp}/host_sanitizers.log" 2>&1 || { tail -30 "${TMPDIR:-/tmp}/host_sanitizers.log"; exit 1; }
tail -1 "${TMPDIR:-/tmp}/host_sanitizers.log"
echo "UBSan reports: $(grep -c 'runtime error' "${TMPDIR:-/tmp}/host_sanitizers.log" || true)"
