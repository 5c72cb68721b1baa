#!/bin/bash
# HISTORY (rounds 4-5): several switches these calls set (SAVGOL_HIP_ROLL_XCD / _TILE_CAP / _BOX, _STREAM_XCD / _TILE / _DMA_GROUP, _MOMENT_FORM, ...) and the round-2 moment objects
# were removed in round 6 (profiles/EXPERIMENTS.md, "Round 6"): the functions below document what produced the r04_* / r05_* files, they no longer all run.
# tools/experiments.sh NAME -- the GPU calls behind profiles/EXPERIMENTS.md, rounds 4 and 5, one function each (rounds 1-3 ran ~60 one-off
# scripts under tools/r3/; their recipes are the command column of profiles/README.md).  Run on the GPU box from the repository root:
#     gpurun --timeout 1500 -- 'bash tools/experiments.sh tile2d_knobs'
# A/B libraries are built beforehand, on the CPU box, with tools/build_variant.sh (the lines marked "build:").
cd "${GRAFT_REPO_ROOT:-.}"
L=savitzky-golay-filter_amd/lib/libsavgol_hip.so
A=tools/ab
N7="-USEP_ROLL_MIN_N -DSEP_ROLL_MIN_N=7"          # build only n = 7 of the group (45 s instead of 90)
clean() { grep -v amdgpu.ids; }

membench_tile2d() {            # R4.1: the bare tile pattern (build: hipcc --offload-arch=gfx950 -O3 -o tools/membench_tile2d tools/membench_tile2d.hip)
    tools/membench_tile2d
}
tile2d_knobs() {               # R4.1  build: for v in r12:-DSG_ROLL_TILE_ROWS=12 r20:... r24:... wpb1:-DSG_ROLL_TILE_WPB=1 wpb4:... skipedge:-DSG_ROLL_SKIP_EDGE_STRIPS; do tools/build_variant.sh ${v%%:*} sg_2d_roll_g1.o "$N7 ${v#*:}"; done
    for images in 64 256; do
        python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 $A/lib_r12.so $A/lib_r20.so $A/lib_r24.so $A/lib_wpb1.so $A/lib_wpb4.so $A/lib_skipedge.so --n 7 --images $images 2>&1 | clean
    done
    python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE_CAP=3 $L@SAVGOL_HIP_ROLL_TILE_CAP=2 $L@SAVGOL_HIP_ROLL_TILE_CAP=1 --n 7 --images 256 2>&1 | clean
    python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_XCD=0 --n 7 --images 256 2>&1 | clean
}
tile2d_edges() {               # R4.1: edge strips on vector loads against the scalar path and the walk, three boundary modes
    for b in 1 0 2; do python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE_EDGE=0 $L@SAVGOL_HIP_ROLL_TILE=0 --n 7 --images 256 --boundary $b 2>&1 | clean; done
    python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n 7 --images 256 --zeros 2>&1 | clean
    for cols in 3840 4080; do python tools/ab_2d.py $L --n 7 --images 256 --cols $cols 2>&1 | clean; done
}
tile2d_other_forms() {         # R4.1: every additive half window, the general one- / two-term forms
    for n in 2 3 4 5 6 7; do python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n $n 2>&1 | clean; done
    for n in 2 4 5 6 7; do for cfg in "3 1 0" "3 2 0" "3 1 1" "4 1 0"; do set -- $cfg
        python tools/ab_2d.py $L $L@SAVGOL_HIP_ROLL_TILE=0 --n $n --order $1 --dx $2 --dy $3 2>&1 | clean; done; done
    python tools/time_2d_derivs.py 2>&1 | grep -E "fused|laplacian"
}
fp32_chains() {                # R4.2  build (once, at the commit before the change): make B=build_old ... -> tools/ab/lib_onechain.so
    python -m pytest tests/test_gpu_1d.py -q -m gpu -k "reference_s_own_fp32_error" -s 2>&1 | grep -E "worst normwise|passed|failed"
    python tools/diag_1d_accuracy.py 5 25 27 31 32 2>&1 | clean
    for n in 32 32 24 20; do python tools/ab_1d.py $L $A/lib_onechain.so --n $n --rounds 30 2>&1 | clean | tail -2; done
}
stream_tile() {                # R4.3  build: tools/build_variant.sh s4r16 sg_stream_roll.o "-DSG_STREAM_TILE_ROWS=16" (s4r24, s2r32, s2r48, s4r16w4 alike; the tile must be enabled for FMA / n = 16 in the variant)
    for T in 4096 8192 16384; do for v in s4r16w4 s4r16 s4r24 s2r32 s2r48; do
        echo "## ticks $T, $v"; TICKS=$T HALF_WINDOWS=4,8,16 SAVGOL_HIP_LIB=$PWD/$A/lib_$v.so python tools/time_stream_block.py 2>&1 | clean; done
        echo "## ticks $T, walk"; TICKS=$T HALF_WINDOWS=4,8,16 SAVGOL_HIP_STREAM_TILE=0 python tools/time_stream_block.py 2>&1 | clean; done
}
strided() {                    # R4.4
    python tools/time_strided.py 2>&1 | clean
}
small_fixes() {                # R4.5: pool repro, pool trim probe, RCCL exchanges, scratch / overlap tests, the 360-point demo
    tools/repro_null_stream_pool; tools/probe_pool_trim
    python -m pytest tests/test_gpu_rccl_exchange.py tests/test_gpu_2d.py tests/test_gpu_1d.py -x -q -m gpu -k "exchange or overlapping or scratch_pool" 2>&1 | tail -3
    savitzky-golay-filter_amd/lib/time_demo360 oracle/_ref/libsavgol_ref.so $L
}
evidence() {                   # the committed r04_* evidence: tools/run_profiles_r4.sh here, then python tools/summarise_profiles_r4.py on the CPU box
    bash tools/run_profiles_r4.sh; IMAGES=256 bash tools/run_profiles_c4_r4.sh
}
# ---- round 5 (tools/r5_run.sh was rewritten per call; these are the calls that produced the committed r05_* files) ----
membench_streamtile() {        # R5.1  build: hipcc --offload-arch=gfx950 -O3 -o tools/membench_streamtile tools/membench_streamtile.hip
    tools/membench_streamtile
}
stream_dma() {                 # R5.1: LDS-DMA tiles against the walk, ring depth / tile height / waves per block / chains (build: make EXTRA=-DSG_DMA_EXPERIMENT ... sg_stream_dma_lo.o into tools/ab/lib_dmaexp.so)
    echo "## default"; HALF_WINDOWS=4,8,16 python tools/time_stream_block.py 2>&1 | clean
    echo "## walk"; SAVGOL_HIP_STREAM_DMA=0 HALF_WINDOWS=4,8,16 python tools/time_stream_block.py 2>&1 | clean
    for cfg in "32 4 8" "32 4 12" "32 4 16" "64 4 12" "48 4 12" "96 4 12" "128 4 12" "64 2 12" "64 8 8"; do set -- $cfg
        echo "## TR=$1 WPB=$2 PAIRS=$3"; SAVGOL_HIP_LIB=$PWD/$A/lib_dmaexp.so SAVGOL_HIP_STREAM_DMA_TR=$1 SAVGOL_HIP_STREAM_DMA_WPB=$2 SAVGOL_HIP_STREAM_DMA_PAIRS=$3 HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | clean; done
    for g in 32 64 256; do echo "## group $g"; SAVGOL_HIP_STREAM_DMA_GROUP=$g HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | clean; done
    echo "## half windows 17-32"; HALF_WINDOWS=17,20,24,32 python tools/time_stream_block.py 2>&1 | clean
}
parity_margins() {             # R5.2: every comparison of the GPU suite with its bar
    rm -f gpurun_out/r5/parity.jsonl; mkdir -p gpurun_out/r5
    SAVGOL_PARITY_LOG=$PWD/gpurun_out/r5/parity.jsonl python -m pytest tests -q -m gpu 2>&1 | tail -3
    python tools/parity_margins.py gpurun_out/r5/parity.jsonl
}
momenth() {                    # R5.7: the half-lane fp32 block moments against round 4's form / the plain sum, one process
    for n in 32 28 24 23 22 20 18 16; do python tools/ab_1d.py $L $L@SAVGOL_HIP_MOMENT_FORM=32 --n $n --rounds 12 2>&1 | clean | tail -2; done
}
stream_skip_taps() {           # R5.8: what the fused tile's multiply-adds cost (build: sg_stream_dma.hip with -DSG_DMA_SKIP_TAPS=k [-DSG_DMA_EXPERIMENT] into tools/ab/lib_skip$k.so / lib_exp$k.so)
    for sk in 3 6 10 14; do SAVGOL_HIP_LIB=$PWD/$A/lib_skip$sk.so HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | clean; done
    for sk in 0 6 14; do for cfg in "4 12" "4 16" "4 24" "4 32" "2 24" "2 32" "8 12" "8 16"; do set -- $cfg
        SAVGOL_HIP_LIB=$PWD/$A/lib_exp$sk.so SAVGOL_HIP_STREAM_DMA_TR=32 SAVGOL_HIP_STREAM_DMA_WPB=$1 SAVGOL_HIP_STREAM_DMA_PAIRS=$2 HALF_WINDOWS=16 python tools/time_stream_block.py 2>&1 | grep "fma=1"; done; done
}
stream_moment() {              # R5.8: block-moment tiles against the tap-by-tap tiles and the walk, interleaved in one process; then process to process
    for n in 12 14 16 20; do python tools/ab_stream.py $L $L@SAVGOL_HIP_STREAM_MOMENT=0 --n $n 2>&1 | clean; done
    python tools/ab_stream.py $L $L@SAVGOL_HIP_STREAM_MOMENT=0 --n 16 --m 2 --d 0 2>&1 | clean
    for i in 1 2 3 4 5 6 7 8; do python tools/ab_stream.py $L $L@SAVGOL_HIP_STREAM_MOMENT=0 $L@SAVGOL_HIP_STREAM_DMA=0 --n 16 --rounds 6 2>&1 | clean; done
}
stream_shapes() {              # R5.8: (waves per block, ring pairs) per half window and bank (build: -DSG_DMA_EXPERIMENT=2 -DSG_DMA_MIN_N=$N -DSG_DMA_MAX_N=$N into tools/ab/lib_few$N.so)
    V="SAVGOL_HIP_STREAM_DMA_TR=32,SAVGOL_HIP_STREAM_DMA_WPB"
    for N in 4 8; do X=$A/lib_few$N.so; for fma in 1 0; do
        python tools/ab_stream.py $X@$V=4,SAVGOL_HIP_STREAM_DMA_PAIRS=12 $X@$V=8,SAVGOL_HIP_STREAM_DMA_PAIRS=12 $X@$V=8,SAVGOL_HIP_STREAM_DMA_PAIRS=16 $X@$V=4,SAVGOL_HIP_STREAM_DMA_PAIRS=16 --n $N --fma $fma 2>&1 | clean; done; done
}
tick_latency() {               # R5.4: per-tick paths from C (launch + synchronise, push_wait, resident service, back-to-back device time)
    savitzky-golay-filter_amd/lib/c_api_demo
}
evidence_r5() {                # the committed r05_* evidence: this, then python tools/summarise_profiles_r5.py on the CPU box
    bash tools/run_profiles_r5.sh
}
"$@"
