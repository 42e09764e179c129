# Debug aid: build libcozk with -DCOZK_COUNT_COPIES (common.hpp counts hipMemcpyAsync / hipMemsetAsync / hipStreamSynchronize per call
# site and prints the table at exit) into co-zkvms_amd/build/libcozk_count.so.  On the GPU box:
#   cp co-zkvms_amd/build/libcozk_count.so co-zkvms_amd/libcozk.so && python tools/run_flow.py --log-n 20 --steps 1 2> copies.err
# (the box works on a scratch copy of the tree; the real libcozk.so here is untouched).  Remove the .so afterwards.
set -e
cd "$(dirname "$0")/../co-zkvms_amd/csrc"
mkdir -p /tmp/cozk_dbgbuild
for f in capi msm poly harness shm_hub ring; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCOZK_COUNT_COPIES -c $f.hip -o /tmp/cozk_dbgbuild/$f.o 2>/dev/null & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/libcozk_count.so /tmp/cozk_dbgbuild/*.o -ldl
ls -la ../build/libcozk_count.so
