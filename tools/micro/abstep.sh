# Same-box A/B of the whole train step between builds of the library: copies tools/micro/_bin/libcclip_hip_<v>.so over the in-tree
# library and runs bench.py, three interleaved rounds (box-to-box variance on the pool is +-2 %, larger than most changes).
# usage (on the GPU box, via gpurun): bash tools/micro/abstep.sh pre nt   -> gpurun_out/ab_<v><i>.log
L=construction-clip_amd/cclip_hip/libcclip_hip.so
cp $L /tmp/libcclip_hip_keep.so
for i in 1 2 3; do
  for v in "$@"; do
    cp tools/micro/_bin/libcclip_hip_$v.so $L
    python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab_$v$i.log 2>&1
    echo "$v $i $(tail -1 gpurun_out/ab_$v$i.log | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
cp /tmp/libcclip_hip_keep.so $L
