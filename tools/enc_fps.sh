#!/bin/bash
# usage: tools/enc_fps.sh [frames] [qp]   (GPU box) - 1080p IPPP encode with the integration-harness encoders:
# reference C++/SSE4 primitives on the CPU vs the same control flow with xMotionSearch served by the GPU library.
# The all-GPU variant (every primitive per call) needs minutes per 1080p picture; it was checked once with
#   python3 tools/gen_yuv.py 1920 1080 2 7 in.yuv; oracle/_ref/x265_gpu_all -i in.yuv -o o.265 -w 1920 -h 1080 -f 2 -q 32 -ip 100 -sceneChange 0 -t 1
#   (md5 fd4a85b8bb2ddafc78756ce8cad577db, 31307 bytes, as oracle/_ref/x265_refsse writes)
frames=${1:-6}; qp=${2:-32}
d=$(mktemp -d)
python3 tools/gen_yuv.py 1920 1080 $frames 7 $d/in.yuv
for v in refsse gpu_me gpu_planes gpu_intra; do
  s=$(date +%s.%N)
  timeout -k 10 600 oracle/_ref/x265_$v -i $d/in.yuv -o $d/$v.265 -w 1920 -h 1080 -f $frames -q $qp -ip 100 -sceneChange 0 -t 1 > $d/$v.log 2>&1 || { echo "$v failed"; tail -3 $d/$v.log; exit 1; }
  e=$(date +%s.%N)
  echo "$v: $(md5sum < $d/$v.265 | cut -c1-32) $(stat -c %s $d/$v.265) bytes, wall $(python3 -c "print(round($e-$s,2))") s, $(grep -o 'Time:.*' $d/$v.log)"
done
rm -rf $d
