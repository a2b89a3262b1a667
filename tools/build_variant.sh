#!/bin/bash
# A measurement build of the library beside the shipped one: tools/build_variant.sh <name> "<extra hipcc flags>"
# -> build/libs4g_hip_<name>.so (objects in build/obj_<name>); used with S4G_HIP_LIB / tools/ab_libs.sh
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build/obj_$name
make -C $root/s4g_release_amd/csrc -j8 OBJDIR=$root/build/obj_$name LIB=$root/build/libs4g_hip_$name.so HIPFLAGS_EXTRA="$*" 2>&1 | grep -v "^/opt/rocm/bin/hipcc" || true
ls -la $root/build/libs4g_hip_$name.so
