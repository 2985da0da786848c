cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "# in-kernel cycle accounting of conv_halo3_kernel (instrumented build, scripts/build_prof_lib.sh + scripts/prof_halo.py)"
echo "# per workgroup (wave 0), s_memtime ticks; the timer reads themselves cost a few per cent"
timeout -k 10 120 python scripts/prof_halo.py 64 112 128 128 10
timeout -k 10 120 python scripts/prof_halo.py 64 56 256 256 10
timeout -k 10 120 python scripts/prof_halo.py 64 112 64 64 10
timeout -k 10 120 python scripts/prof_halo.py 64 224 32 32 10
} > gpurun_out/halo_cycle_profile.log 2>&1
cat gpurun_out/halo_cycle_profile.log | grep -v amdgpu.ids
