#!/bin/bash
# The command behind every file under profiles/ (run on an MI355X box; from the development container prefix a line with
# `gpurun --timeout 1800 -- '...'`).  `scripts/reproduce_profiles.sh list` prints the table, `scripts/reproduce_profiles.sh
# <file>` runs the command(s) of one file and leaves the raw output under gpurun_out/ (the files under profiles/ are the
# summaries that were kept, some with hand-written headers).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
tools() { make -C mpifft4py_amd/csrc -j8 >/dev/null && make -C tools "$@" >/dev/null; }
run() {
  case "$1" in
    r01_final_*|r01_v0_*) echo "(round-1 builds: git checkout the round-1 tag, then) scripts/profile_cmd.sh r01 bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off" ;;
    r01_kbench_variants.txt) tools kbench; tools/build/kbench ;;
    r01_kbench2_long_and_fp32.txt) tools kbench2; tools/build/kbench2 ;;
    r01_size_sweep.txt|r02_size_sweep.txt|r03_size_sweep.txt) bash scripts/size_sweep.sh ;;
    r03_kbench3_occ512.txt) tools kbench3; tools/build/kbench3 occ512 5; tools/build/kbench3 occ256 5 ;;
    r02_final_*) bash scripts/profile_r02.sh bench; python scripts/summarize_profiles.py r02_final gpurun_out/prof_r02/trace gpurun_out/prof_r02/fetch gpurun_out/prof_r02/write "bench.py 1024^3 fp64"; python scripts/summarize_profiles.py sq r02_final gpurun_out/prof_r02/sq1 gpurun_out/prof_r02/sq2 ;;
    r02_pack_*) bash scripts/profile_r02.sh pack; python scripts/summarize_profiles.py r02_pack gpurun_out/prof_r02/pack_trace gpurun_out/prof_r02/pack_fetch gpurun_out/prof_r02/pack_write "scripts/pack_workload.py 512" ;;
    r02_mask_*) bash scripts/profile_cmd.sh mask scripts/maskprof.py 1024 double; python scripts/summarize_profiles.py r02_mask gpurun_out/prof_mask/trace gpurun_out/prof_mask/fetch gpurun_out/prof_mask/write "scripts/maskprof.py 1024 double" ;;
    r02_membench_yardsticks.txt|r02_membench_stamped.txt|r02_infinity_cache_probe.txt) tools membench; tools/build/membench; tools/build/membench stamp; tools/build/membench mall ;;
    r02_kbench3_variants.txt|r02_colfft_experiments.md) tools kbench3; tools/build/kbench3 "" 5; tools/build/kbench3 persist 5 ;;
    r02_kbench3_long_lengths.txt) tools kbench3 membench; tools/build/membench tilelong; for f in long twl occ small f32long half f32k2 h1536; do tools/build/kbench3 $f 3; done ;;
    r02_power_of_two_stride.txt) tools kbench3; KB_PADSWEEP=1 tools/build/kbench3 padplane 3; KB_STRIDESWEEP=1 tools/build/kbench3 padplane 3; python scripts/c2cprof.py 1024 double; python scripts/c2cprof.py 1024 single; python scripts/c2cprof.py 2048 single ;;
    r02_cache_fusion_sweeps.txt) echo "(experimental plan paths, removed after the measurement: see r02_colfft_experiments.md)" ;;
    r02_c2c_stage_times.txt) python scripts/c2cprof.py 1024 double; python scripts/c2cprof.py 2048 single ;;
    r02_row_kernels_long_lengths.txt) for n in 1152 1280 1536; do python scripts/r2cprof.py $n double; done; python scripts/padprof.py 1024 slab ;;
    r02_pencil_zfuse_512.txt) python scripts/pencil_prof.py 512 8; MFFT_NO_ZFUSE=1 python scripts/pencil_prof.py 512 8 ;;
    r02_run_to_run_spread.txt) bash scripts/bimodal_r02.sh ;;
    r02_placement_probe.txt) python scripts/placement_probe.py ;;
    r02_placement_reroll.txt) python scripts/reroll_probe.py ;;
    r02_two_thirds_rule_mask.txt) python scripts/maskprof.py 1024 double; python scripts/maskprof.py 1024 single; python scripts/maskprof_ranks.py 1024 2; python scripts/maskprof_ranks.py 1024 8 ;;
    r02_ipc_fanout_probe.txt)   # the per-peer-stream flag form was removed in round 4: the old revision is built in a throw-away worktree
      # (its own sources, objects and library; this tree and its libmpifft4py_amd.so are not touched)
      wt=$(mktemp -d /tmp/mfft-r02-fanout.XXXXXX) && git worktree add --detach "$wt" 292508d >/dev/null &&
      make -C "$wt/mpifft4py_amd/csrc" -j8 >/dev/null &&
      (cd "$wt" && MFFT_IPC_PULL=streams MFFT_IPC_STREAM_FLAGS=1 python bench.py --gpus 8 --size 128 --steps 3 --warmup 1 --cpu-baseline off --pencil-extra off --transport ipc);
      git worktree remove --force "$wt" 2>/dev/null ;;
    r02_bench_after_two_wg_plan.json|r02_final_bench_1024cubed.json|r03_final_bench_1024cubed.json) python bench.py --steps 10 --warmup 3 ;;
    r04_final_bench_1024cubed.json) python bench.py --steps 10 --warmup 3 ;;
    r04_final_*) bash scripts/profile_r04.sh bench; python scripts/summarize_profiles.py r04_final gpurun_out/prof_r04/trace gpurun_out/prof_r04/fetch gpurun_out/prof_r04/write "bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off: 1024^3 fp64 slab R2C forward+inverse on one MI355X"; python scripts/summarize_profiles.py sq r04_final gpurun_out/prof_r04/sq1 gpurun_out/prof_r04/sq2 ;;
    r04_720_*) bash scripts/profile_cmd.sh b720 bench.py --size 720 --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off; python scripts/summarize_profiles.py r04_720 gpurun_out/prof_b720/trace gpurun_out/prof_b720/fetch gpurun_out/prof_b720/write "bench.py --size 720: 720^3 fp64 slab R2C forward+inverse on one MI355X" ;;
    r04_radix7_sweep.txt) bash scripts/archive/r04_gpu1.sh ;;
    r04_wave_packed_real_kernels.txt) echo '(build with every candidate wave-packed: registry.h wave_packable without the E % 15 / thread-count conditions)'; bash scripts/archive/r04_gpu4.sh ;;
    r04_xpass_ab.txt) bash scripts/archive/r04_gpu3.sh ;;
    r04_xpass_kernel_ab.txt) python scripts/xpass_kernel_ab.py ;;
    r04_xpass_stride_map.txt) (cd scripts && python xpass_stride_map.py 1024 && python xpass_stride_map.py 2048 && python xpass_pad_sweep.py) ;;
    r04_comm_priority.txt) bash scripts/archive/r04_priority.sh ;;
    r04_size_sweep.txt) bash scripts/size_sweep.sh ;;
    r05_size_sweep.txt) bash scripts/archive/r05_final.sh gate ;;
    r05_final_*) bash scripts/archive/r05_final.sh prof ;;
    r05_radix7_c2r_bisect.txt|r05_r03_vs_head_ab.txt) echo "(needs reduced libraries of old commits under _bisect/: see the header of scripts/archive/r05_gpu1.sh)" ;;
    r05_alloc_shift_probe.txt) bash scripts/archive/r05_gpu2.sh ;;
    r05_pad_align_ab.txt|r05_pad_align_bits.txt) bash scripts/archive/r05_gpu3.sh; bash scripts/archive/r05_gpu4.sh ;;
    r05_config5_zpitch.txt) python scripts/config5_full.py; MFFT_NO_ZPITCH=1 python scripts/config5_full.py ;;
    r05_pad_pmc_traffic.txt) bash scripts/pad_pmc_r05.sh ;;
    r05_radix42_sweep.txt) bash scripts/archive/r05_gpu13.sh ;;
    r05_ytile_builds.txt) make -C tools kbench3 membench; for f in tw64 y64 y64b tw1536; do tools/build/kbench3 $f 3; done; tools/build/membench tile1024w; bash scripts/archive/r05_gpu25.sh; bash scripts/archive/r05_gpu28.sh ;;
    r05_plain_rows.txt) echo "git checkout <the commit 'plain rows: measured'>; make -C mpifft4py_amd/csrc -j8 && make -C tools kbench3; bash scripts/archive/r05_gpu16.sh" ;;
    r05_wave_placement.txt) make -C tools occ_probe membench kbench3; tools/build/occ_probe; tools/build/membench stamp1200; tools/build/membench tile1200; tools/build/kbench3 occ1200 5; tools/build/kbench3 narrow 3; tools/build/kbench3 wide16 5; tools/build/kbench3 wideb 5; bash scripts/archive/r05_gpu19.sh; bash scripts/archive/r05_gpu26.sh ;;
    r05_miscompile_cure_modes.txt) make -C tools rowcheck2_0 rowcheck2_1 rowcheck2_2 rowcheck2_3 rowcheck2_4; for m in 0 1 2 3 4; do tools/build/rowcheck2_$m; done ;;
    r06_nlz_variants.txt) echo "(experiment build: make -C mpifft4py_amd/csrc CXXFLAGS='... -DMFFT_NLZ_EXPERIMENTS' at the commit named in the file, then) bash scripts/archive/r06_gpu2.sh; bash scripts/archive/r06_gpu3.sh" ;;
    r06_nlz_pmc_counters.txt) bash scripts/archive/r06_gpu4.sh ;;
    r06_nlz_wave.txt) bash scripts/archive/r06_gpu11.sh ;;
    r06_dns_batch.txt) bash scripts/archive/r06_gpu5.sh ;;
    r06_dns_512_kernel_stats.csv|r06_dns_1024.txt) bash scripts/archive/r06_gpu7.sh ;;
    r06_placement_pmc.txt) bash scripts/archive/r06_placement1.sh; bash scripts/archive/r06_placement2.sh ;;
    r06_perf_gate.log|r06_size_sweep.txt) bash scripts/archive/r06_gate.sh ;;
    r06_pitched_spectrum.txt) bash scripts/archive/r06_gpu9.sh; bash scripts/archive/r06_gpu10.sh ;;
    r06_any_n_sweep.txt) bash scripts/archive/r06_anyn.sh ;;
    r06_radix70.txt) bash scripts/archive/r06_gpu19.sh ;;
    r06_final_*|r06_dns_kernel_stats.csv|r06_dns_pmc_traffic.json) bash scripts/r06_final.sh prof ;;
    r06_bench_dev_run.json|r06_bench_2ranks_ipc.json|r06_bench_2ranks_mock.json) bash scripts/r06_final.sh bench ;;
    r06_576_kernel_stats.csv|r06_576_pmc_traffic.json) bash scripts/archive/r06_prof576.sh ;;
    r06_group_t.txt) echo "(before / after: git archive <commit before group T> into _ab/prev, make there, then)"; bash scripts/archive/r06_group_t_ab.sh ;;
    r06_group_uv.txt) echo "(before / after: git archive <commit before groups U, V> into _ab/prev, make there, then)"; bash scripts/archive/r06_group_uv_ab.sh ;;
    r06_c2r_mlds.txt) bash scripts/archive/r06_gpu14.sh; bash scripts/archive/r06_mlds_t.sh; bash scripts/archive/r06_mlds_long.sh; bash scripts/archive/r06_mlds_e20.sh ;;
    r06_dns_23rule.txt) bash scripts/archive/r06_dns23.sh; echo "(before: MFFT_NO_PRUNE=1)"; python scripts/maskprof.py 1024 double pitched ;;
    r05_col_occupancy_caps.txt|r05_row_occupancy_caps.txt) echo "(needs the library without the caps under _ab/old: see scripts/archive/r05_gpu7.sh / r05_gpu8.sh)" ;;
    r04_rank_shapes.txt) python scripts/rank_shapes.py ;;
    r04_ypass_pitch.txt) python scripts/ypass_pitch_ab.py ;;
    r04_col3_1536.txt) bash scripts/archive/r04_col3.sh ;;
    r04_aligned_route_ab.txt) echo '(the switch MFFT_ALIGNED exists up to commit a7fb791 only)'; bash scripts/aligned_ab.sh ;;
    r04_p1_xpad_ab.txt) bash scripts/p1_xpad_ab.sh ;;
    r04_fwd_oop_ab.txt) echo '(MFFT_FWD_OOP=2 exists up to commit a7fb791 only)'; bash scripts/fwd_oop_ab.sh ;;
    r04_small_mesh_overhead.txt) python scripts/small_mesh_overhead.py ;;
    r04_ipc_soak.txt) bash scripts/archive/r04_soak.sh ;;
    r04_col3s_ab.txt) bash scripts/col3s_ab.sh ;;
    r04_pad_pmc_traffic.txt) bash scripts/pad_pmc.sh ;;
    r04_serialised_loads.txt) echo '(per translation unit: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --offload-device-only mpifft4py_amd/csrc/kernels_X_Y.hip -o /tmp/X_Y.s; python scripts/isa_scan.py /tmp/*.s)'; bash scripts/col3s_f32.sh; bash scripts/archive/r04_c2r_check.sh ;;
    r03_final_*) bash scripts/profile_r03.sh bench; python scripts/summarize_profiles.py r03_final gpurun_out/prof_r03/trace gpurun_out/prof_r03/fetch gpurun_out/prof_r03/write "bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off: 1024^3 fp64 slab R2C forward+inverse on one MI355X"; python scripts/summarize_profiles.py sq r03_final gpurun_out/prof_r03/sq1 gpurun_out/prof_r03/sq2 ;;
    r03_720_*) bash scripts/profile_cmd.sh b720 bench.py --size 720 --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off; python scripts/summarize_profiles.py r03_720 gpurun_out/prof_b720/trace gpurun_out/prof_b720/fetch gpurun_out/prof_b720/write "bench.py --size 720: 720^3 fp64 slab R2C forward+inverse on one MI355X"; python scripts/summarize_profiles.py sq r03_720 gpurun_out/prof_b720/sq1 gpurun_out/prof_b720/sq2 ;;
    r03_512_*) bash scripts/profile_cmd.sh b512 bench.py --size 512 --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off; python scripts/summarize_profiles.py r03_512 gpurun_out/prof_b512/trace gpurun_out/prof_b512/fetch gpurun_out/prof_b512/write "bench.py --size 512: 512^3 fp64 slab R2C forward+inverse on one MI355X"; python scripts/summarize_profiles.py sq r03_512 gpurun_out/prof_b512/sq1 gpurun_out/prof_b512/sq2 ;;
    r03_mixed_radix_15.txt) tools rowcheck; tools rowcheck_nolaunder; bash scripts/archive/r03_plans15.sh ;;
    r03_composite_radix.txt) for n in 288 400 500 576 640 800 1000 1152 1280 1600; do python bench.py --size $n --steps 5 --cpu-baseline off --pencil-extra off 2>/dev/null | python scripts/show_bench.py | head -1; done; for n in 576 800 1000 1152 1280 1600; do python bench.py --size $n --precision single --steps 5 --cpu-baseline off --pencil-extra off 2>/dev/null | python scripts/show_bench.py | head -1; done; echo '(old library: git checkout a8df077 -- mpifft4py_amd/csrc/plans.h, rebuild, rerun)' ;;
    r03_kbench3_quarter_exchange.txt) tools kbench3; tools/build/kbench3 q1536 5 ;;
    r03_cu_mask_probe.txt) tools overlap_probe; tools/build/overlap_probe 8 16 32 ;;
    r03_overlap.txt) for c in "p4_kz4 4 1024 4 1" "p4_rows4 4 1024 -4 1" "p2_kz4 2 1024 4 1" "p4_kz4_copy 4 1024 4 0" "p8_kz4 8 1024 4 1"; do scripts/overlap_trace.sh $c; set -- $c; python scripts/summarize_overlap.py gpurun_out/overlap_$1; done ;;
    r03_ipc_pull_modes.txt) bash scripts/archive/r03_first_gpu.sh; bash scripts/archive/r03_gpu3.sh; MFFT_IPC_STREAM_FLAGS=1 python scripts/ipc_stress.py 8 2 -4 60 128 ;;   # (the switch exists up to commit 292508d only)
    r03_shared_gpu_pipeline_latency.txt) bash scripts/archive/r03_gpu2.sh ;;
    r03_two_thirds_rule_ranks.txt) python scripts/maskprof_ranks.py 1024 8 ;;
    r03_pencil_dealias.txt) for k in X Y; do for p in double single; do python scripts/maskprof.py 1024 $p $k; MFFT_NO_PRUNE=1 python scripts/maskprof.py 1024 $p $k; done; python scripts/padprof.py 512 $k; done; python scripts/padprof.py 512 slab ;;
    *) echo "no recipe for $1" >&2; return 1 ;;
  esac
}
if [ "${1:-list}" = "list" ]; then
  for f in profiles/r0*; do
    b=$(basename "$f")
    printf "%-44s %s\n" "$b" "$(grep -F -m1 "$(echo "$b" | sed 's/_final_.*/_final_*/; s/_pack_.*/_pack_*/; s/^\(r03_[0-9]*\)_.*/\1_*/; s/_mask_[a-z_]*\.\(csv\|json\)/_mask_*/')" "$0" | sed 's/^ *[^)]*) *//; s/ ;;$//' | cut -c1-150)"
  done
  exit 0
fi
run "$1"
