# phase stamps of fps_pruned_kernel on the GPU box: debug build into gpurun_out/, then tools/fps_stamps.py
set -e
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/stamps_build; mkdir -p $O
make -s -C $R/s4g_release_amd/csrc OBJDIR=$O LIB=$O/libs4g_hip_stamps.so HIPFLAGS_EXTRA=-DS4G_FPS_STAMPS -j8 > $O/build.log 2>&1
S4G_HIP_LIB=$O/libs4g_hip_stamps.so python3 $R/tools/fps_stamps.py ${1:-16}
S4G_HIP_LIB=$O/libs4g_hip_stamps.so python3 $R/tools/fps_stamps.py 1
rm -rf $O
