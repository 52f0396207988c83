# usage (GPU box): bash tools/exp_stream_sweep.sh [C4|C5] -- the stream renderer by waves-per-SIMD build and issue thresholds
# (RR_DEBUG_ASYNC="step,shade" in sixteenths of the wave's live lanes: a step is issued once that many lanes wait for it -- the
# node loop needs twice as many --, a shading pass once shade/16 of them are finished)
mkdir -p gpurun_out
which=${1:-C5}
for w in 6; do for a in 2,6 1,6 1,4 2,4 3,6 2,8 3,8 1,8; do echo "waves $w async $a: $(RR_DEBUG_KERNEL=stream RR_DEBUG_ASYNC=$a timeout -k 10 100 python tools/exp_tlas.py $which 2 2>&1 | tail -1 | cut -c1-60,150-330)"; done; done
