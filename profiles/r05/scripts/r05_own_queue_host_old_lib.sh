#!/bin/bash
# the C++ own-queue host against a library from BEFORE the origins upload was taken off the own-queue stream (tools/_ab/libvtmc_packed3.so): does the
# pinned copy on a CU-mask stream hang this pattern too?  (timeout 60 s; the hang, if any, is in the runtime's tear-down: nothing is left running)
mkdir -p /tmp/oq_old && cp tools/_ab/libvtmc_packed3.so /tmp/oq_old/libvtmc.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -I include -o /tmp/oq_old/host tools/calib/own_queue_host.hip -L /tmp/oq_old -lvtmc -Wl,-rpath,/tmp/oq_old 2>/dev/null || exit 1
timeout -k 5 60 /tmp/oq_old/host; echo "old library: rc=$?"
