# usage (GPU box): bash tools/exp_stream_sweep.sh [C4|C5] -- the stream renderer by waves-per-SIMD build and vote thresholds
# (RR_DEBUG_ASYNC="node,shade" in eighths: the node loop keeps going while more than (8 - node)/8 of its lanes still descend; a
# shading pass needs shade/8 of the wave finished)
mkdir -p gpurun_out
which=${1:-C5}
for w in 6; do for a in 1,2 1,3 2,2 2,3 2,4 3,3 3,4 4,4; do echo "waves $w async $a: $(RR_DEBUG_KERNEL=stream RR_DEBUG_STREAM_WAVES=$w RR_DEBUG_ASYNC=$a timeout -k 10 100 python tools/exp_tlas.py $which 2 2>&1 | tail -1 | cut -c1-60,150-330)"; done; done
