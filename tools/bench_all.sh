# every bench.py mode quoted in DESIGN.md section 6, one after the other (run on the GPU box via gpurun)
set -e
o=gpurun_out/final
mkdir -p $o
python bench.py > $o/default.log 2>&1
python bench.py --mode fwd --no-cpu-baseline > $o/fwd.log 2>&1
python bench.py --mode image --steps 20 --warmup 5 > $o/image.log 2>&1
python bench.py --mode caption > $o/caption.log 2>&1
python bench.py --mode image --model ViT-L/14@336px --batch 256 --dtype bf16 --no-cpu-baseline > $o/l14_bf16.log 2>&1
python bench.py --mode image --model ViT-L/14@336px --batch 256 --dtype fp8 --no-cpu-baseline > $o/l14_fp8.log 2>&1
python bench.py --mode image --dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline > $o/image_fp8.log 2>&1
python bench.py --tower-streams 1 --no-cpu-baseline > $o/single.log 2>&1
python tools/decode_bench.py > $o/decode.log 2>&1
for f in default fwd image caption l14_bf16 l14_fp8 image_fp8 single; do echo "== $f"; tail -1 $o/$f.log | cut -c1-260; done
tail -2 $o/decode.log
