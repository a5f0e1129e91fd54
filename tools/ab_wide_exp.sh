#!/bin/bash
# GPU box: attention tests that touch the wide forward, then the wide forward against another build of the
# library (my_flash_attn_cuda/exp_<name>.so, default "prev": python flash-attention_amd/build.py --exp prev)
name=${1:-prev}
mkdir -p gpurun_out/wide
timeout -k 10 900 python -u -m pytest tests/test_attention_gpu.py tests/test_attention_causal_gpu.py tests/test_fuzz_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/wide/tests.log 2>&1
rc=$?; tail -4 gpurun_out/wide/tests.log; [ $rc -ne 0 ] && exit $rc
OLD=flash-attention_amd/my_flash_attn_cuda/exp_$name.so
for sh in 4,16,4096,256 4,16,4096,192 2,8,8192,256; do
  echo "== $sh"; timeout -k 10 200 python tools/ab_attn.py "attn_variant=-1" "attn_variant=-1,lib=$OLD" --shape $sh || exit 1
done
echo "== causal 4,16,4096,256"; timeout -k 10 200 python tools/ab_attn.py "attn_variant=-1" "attn_variant=-1,lib=$OLD" --shape 4,16,4096,256 --causal || exit 1
echo "== ones"; timeout -k 10 200 python tools/ab_attn.py "attn_variant=-1" "attn_variant=-1,lib=$OLD" --shape 4,16,4096,256 --data ones
