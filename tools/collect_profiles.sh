#!/bin/bash
# Everything under profiles/rNN_* from ONE gpurun call (run from the repository root on the GPU box):
#     gpurun -- "OMG_GIT_HEAD=$(git rev-parse HEAD) bash tools/collect_profiles.sh r06"
# (the variable goes INSIDE the command: gpurun does not forward the caller's environment, and the box has no .git)
# writes gpurun_out/fin/<prefix>_*; copy what is to be judged into profiles/.
p=${1:-r06}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/fin
rm -rf "$out"; mkdir -p "$out"
cd "$root"
last() { tail -1 "$1" > "$2"; }
# the PMC passes first: the bench line quotes their traffic (profiles/${p}_pmc_*.json) while the kernel sources hash to it
timeout 300 python bench.py --no-cpu --no-plain --no-lex --no-sets --no-dropin --no-config4 --no-config1 > $out/quick.log 2>/dev/null; last $out/quick.log $out/quick.json
PASSES="1 2 4 5 6 7" bash tools/pmc_plane.sh gpurun_out/fin/pmc > $out/pmc.log 2>&1
python tools/pmc_any.py gpurun_out/fin/pmc > $out/${p}_pmc_plane_kernels.txt 2>&1
python tools/pmc_plane_json.py gpurun_out/fin/pmc $out/quick.json $out/${p}_pmc_plane_down.json \
  "tools/pmc_plane.sh passes 6 and 7 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-include-regex plane_kernel, tools/prof_cycle.py --steps 2), table in profiles/${p}_pmc_plane_kernels.txt" > $out/pmc_json.log 2>&1
cp $out/${p}_pmc_plane_down.json $root/profiles/${p}_pmc_plane_down.json
python3 tools/prof_config4.py --steps 1 > /dev/null 2>&1          # (generates and caches the configs[4] operator under /tmp/cfg4)
bash tools/pmc_s27.sh gpurun_out/fin/pmc27 > $out/pmc27.log 2>&1
python tools/pmc_any.py gpurun_out/fin/pmc27 > $out/${p}_pmc_s27_kernels.txt 2>&1
python tools/pmc_s27_json.py gpurun_out/fin/pmc27 $out/${p}_pmc_s27_sweep.json \
  "tools/pmc_s27.sh passes 5 and 6 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-include-regex s27_, tools/prof_config4.py --steps 2 --batched 0), table in profiles/${p}_pmc_s27_kernels.txt" > $out/pmc27_json.log 2>&1
cp $out/${p}_pmc_s27_sweep.json $root/profiles/${p}_pmc_s27_sweep.json
timeout 900 python bench.py > $out/bench.log 2>$out/bench.err; last $out/bench.log $out/${p}_bench.json
timeout 300 python bench.py --graph 1 --no-cpu --no-plain --no-lex --no-sets --no-dropin --no-config4 --no-config1 > $out/graph.log 2>/dev/null; last $out/graph.log $out/${p}_bench_hipgraph.json
timeout 300 python bench.py --no-cpu --no-plain --no-lex --no-dropin --no-config4 --no-config1 --dtype f32 > $out/f32.log 2>/dev/null; last $out/f32.log $out/${p}_bench_f32.json
OMG_DIST_P2P=0 timeout 300 python bench.py --dist 1 --no-cpu > $out/dist1.log 2>/dev/null; last $out/dist1.log $out/${p}_bench_dist1.json
OMG_DIST_P2P=1 timeout 300 python bench.py --dist 1 --no-cpu > $out/dist1p.log 2>/dev/null; last $out/dist1p.log $out/${p}_bench_dist1_peer.json
PYTHONPATH=$root timeout 300 python tools/exchange_probe.py > $out/${p}_exchange_probe.txt 2>/dev/null
( for m in 0 1; do echo "OMG_LOOPBACK_P2P=$m (0: device copies in place of RCCL, 1: peer stores between the slabs)"; OMG_LOOPBACK_P2P=$m PYTHONPATH=$root timeout 300 python tools/pdist_loopback_time.py 1 2 4 8 2>/dev/null | grep world; done ) > $out/${p}_slab_loopback.txt
timeout 600 python tools/run_configs.py > $out/${p}_configs.txt 2>&1
timeout 900 python tools/config3_single.py 512 6 > $out/${p}_config3_single_gpu.txt 2>&1
timeout 600 python tools/config4_probe.py --size 256 --cache /tmp/cfg4 2>&1 | tail -1 > $out/${p}_config4_256_fp32.txt
OMG_MARCH_DEBUG=1 timeout 120 python tools/march_probe.py 8x8x2048 8x64x2048 32x32x32 128x128x128 256x256x256 2> $out/${p}_march_timeline.txt > /dev/null
OMG_SETUP_TIMING=1 timeout 120 python tools/setup_timing.py > $out/${p}_setup_timing.txt 2>&1
OMG_SETUP_TIMING=1 timeout 200 python tools/setup_breakdown.py > $out/${p}_setup_breakdown.txt 2>&1
python tools/hostmem_probe.py > $out/${p}_hostmem_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o b -- python3 $root/bench.py --no-cpu > $out/under.log 2>/dev/null
last $out/under.log $out/${p}_bench_under_rocprof.json
cp $out/trace/b_kernel_stats.csv $out/${p}_bench_kernel_stats.csv
python3 $root/tools/trace_stats.py $out/trace/b_kernel_trace.csv > $out/${p}_bench_kernel_stats_by_grid.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/cyc -o c -- python3 $root/tools/prof_cycle.py --steps 6 > /dev/null 2>&1
python3 $root/tools/cycle_timeline.py $out/cyc/c_kernel_trace.csv -3 > $out/${p}_bench_cycle_timeline.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/c4b -o t -- python3 $root/tools/prof_config4.py --steps 4 --batched 1 > /dev/null 2>&1
python3 $root/tools/trace_dump.py $out/c4b/t_kernel_trace.csv 75 > $out/${p}_config4_cycle_timeline_batched.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/c4s -o t -- python3 $root/tools/prof_config4.py --steps 4 --batched 0 > /dev/null 2>&1
python3 $root/tools/trace_dump.py $out/c4s/t_kernel_trace.csv 60 > $out/${p}_config4_cycle_timeline.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/lex -o l -- python3 $root/tools/run_configs.py 0 4 5 > /dev/null 2>&1
python3 $root/tools/march_trace.py $out/lex/l_kernel_trace.csv > $out/${p}_march_by_level.txt 2>&1
# round 6: the opt-in line-scan sweep beside the wavefront kernel, per level of the 256^3 problem (V(1,1) cycles with each), and
# whether its results agree
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/scan -o s -- python3 $root/tools/scan_probe.py time 256 > $out/scan_time.txt 2>/dev/null
( grep "grids scan" $out/scan_time.txt; python3 $root/tools/scan_trace.py $out/scan/s_kernel_trace.csv ) > $out/${p}_scan_by_level.txt 2>&1
for m in 0 1; do PYTHONPATH=$root timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/peer$m -o p -- python3 $root/tools/prof_pdist.py $m 20 > /dev/null 2>&1; done
python3 $root/tools/peer_mode_table.py $out/peer0 $out/peer1 > $out/${p}_peer_mode.txt 2>&1
python3 $root/tools/level_times.py $out/cyc/c_kernel_trace.csv > $out/${p}_level_times.txt 2>&1
cd "$root"
# round 5: configs[4] through the multi-GPU code path with one rank (27-point slabs, omg_sdist), the matrix-free SpMV and where
# it writes, configs[0], update_fine, one slab against the single-GPU hierarchy
timeout 600 python bench.py --dist 1 --stencil 27var --dtype f32 --steps 20 --repeats 5 2>/dev/null | tail -1 > $out/${p}_bench_dist1_27var_f32.json
PYTHONPATH=$root timeout 300 python tools/spmv_place.py 2>&1 | tail -1 > $out/${p}_spmv_destinations.txt
timeout 300 python tools/spmv_probe.py 256 2>&1 > $out/${p}_spmv_probe.txt; OMG_PLANE_SPMV=0 timeout 300 python tools/spmv_probe.py 256 >> $out/${p}_spmv_probe.txt 2>&1
OMG_SETUP_TIMING=1 timeout 600 python tools/update_probe.py 256 5 2>&1 | grep -E "update|norm" | tail -12 > $out/${p}_update_probe.txt
timeout 900 python tools/slab27_ab.py 256 3 2>&1 | tail -4 > $out/${p}_slab27_vs_single.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/upd -o u -- python3 $root/tools/update_probe.py 256 5 > /dev/null 2>&1
grep -E "Name|gjb_|s27_rap|s27_build|extract_inverse|copy_block|narrow_kernel|fill_aug|csr_scatter" $out/upd/u_kernel_stats.csv > $out/${p}_update_kernel_stats.csv
cd "$root"
# round 6: the 7-point per-row-coefficient passes (var7.hip), the coarse solve beyond the old limits, the N > 1 RCCL call sites
# with 8 / 2 rank processes on this GPU (tests/fake_rccl: rehearsals, not scaling measurements)
timeout 300 python tools/var7_probe.py 256 2>&1 | tail -4 > $out/${p}_var7.txt
timeout 600 python tools/coarse_chain_probe.py 64 2>&1 | tail -5 > $out/${p}_coarse_chain.txt
make -C tests/fake_rccl > /dev/null 2>&1
OMG_DIST_SHARED_GPU=rccl OMG_RCCL_LIB=$root/tests/fake_rccl/libfake_rccl.so timeout 900 python bench.py --gpus 8 --no-cpu --size 128 --steps 10 --warmup 2 --repeats 3 2>/dev/null | tail -1 > $out/${p}_bench_gpus8_rehearsal_shared_gpu.json
OMG_DIST_SHARED_GPU=rccl OMG_RCCL_LIB=$root/tests/fake_rccl/libfake_rccl.so timeout 900 python bench.py --gpus 2 --stencil 27var --dtype f32 --no-cpu --size 64 --steps 10 --warmup 2 --repeats 3 2>/dev/null | tail -1 > $out/${p}_bench_gpus2_27var_rehearsal_shared_gpu.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/v7 -o v -- python3 $root/tools/var7_probe.py 256 > /dev/null 2>&1
grep -E "Name|var7_pass" $out/v7/v_kernel_stats.csv > $out/${p}_var7_kernel_stats.csv
cd "$root"
rm -rf $out/trace $out/v7 $out/cyc $out/lex $out/pmc $out/pmc27 $out/peer0 $out/peer1 $out/c4b $out/c4s $out/upd
ls -la $out
