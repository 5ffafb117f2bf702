#!/usr/bin/env python3
"""gpurun_out/ of the last tools/gpu_validate.sh run -> profiles/ (trimmed rocprofv3 kernel stats, bench JSON lines, PMC summaries)."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
RND = sys.argv[1] if len(sys.argv) > 1 else 'r06'      # the round the summaries are named for


def newest(pat):
    fs = glob.glob(pat)
    fs.sort(key=os.path.getmtime)
    return fs[-1]


def trim(src, dst):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if r['Name'].startswith('sdv_') or r['Name'].startswith('__amd_rocclr')]
    with open(dst, 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in keep:
            w.writerow(r)


trim(newest('gpurun_out/prof_bench/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_kernel_stats.csv')
trim(newest('gpurun_out/prof_stitch/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_stitch_kernel_stats.csv')
if glob.glob('gpurun_out/prof_headline/*/*kernel_stats.csv'):
    trim(newest('gpurun_out/prof_headline/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_headline_kernel_stats.csv')
    shutil.copy('gpurun_out/prof_headline.json', f'profiles/{RND}_rocprofv3_headline_bench_line.json')
trim(newest('gpurun_out/prof_pcm1/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_pcm1_kernel_stats.csv')
for d, name in (('prof_p1f', 'pcm1_frames'), ('prof_p16f', 'pcm16x0_frames'), ('prof_p16s', 'pcm16x0_stitch'), ('prof_audio', 'audio')):
    if glob.glob(f'gpurun_out/{d}/*/*kernel_stats.csv'):
        trim(newest(f'gpurun_out/{d}/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_{name}_kernel_stats.csv')
if glob.glob('gpurun_out/prof_pcm1f/*/*kernel_stats.csv'):
    trim(newest('gpurun_out/prof_pcm1f/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_pcm1_front_kernel_stats.csv')
    subprocess.check_call([sys.executable, 'tools/pmc_to_json.py', 'sdv_k_pcm1_lines', f'profiles/{RND}_pmc_sdv_k_pcm1_lines.json', 'p1fpmc'], stdout=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, 'tools/pmc_to_json.py', 'sdv_k_pcm1_lines_lean', f'profiles/{RND}_pmc_sdv_k_pcm1_lines_lean.json', 'p1fpmc'], stdout=subprocess.DEVNULL)
shutil.copy('gpurun_out/bench_full.json', f'profiles/{RND}_bench_full.json')       # every leg's full object (bench.py's gpurun_out/bench_details.json)
if os.path.exists('gpurun_out/bench_line.json'): shutil.copy('gpurun_out/bench_line.json', f'profiles/{RND}_bench_line.json')      # the one line the driver reads
shutil.copy('gpurun_out/bench_2rank_gloo.json', f'profiles/{RND}_bench_2rank_gloo_one_gpu.json')
if os.path.exists('gpurun_out/bench_nccl_1rank.json') and os.path.getsize('gpurun_out/bench_nccl_1rank.json') > 100:
    shutil.copy('gpurun_out/bench_nccl_1rank.json', f'profiles/{RND}_bench_nccl_1rank.json')
subprocess.check_call([sys.executable, 'tools/pmc_to_json.py', 'sdv_k_stc007_frames_lean', f'profiles/{RND}_pmc_sdv_k_stc007_frames_lean.json', 'pmc'], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, 'tools/pmc_to_json.py', 'sdv_k_pcm1_frames', f'profiles/{RND}_pmc_sdv_k_pcm1_frames.json', 'p1pmc'], stdout=subprocess.DEVNULL)
if glob.glob('gpurun_out/apmc3/**/*_counter_collection.csv', recursive=True):
    subprocess.check_call([sys.executable, 'tools/pmc_to_json.py', 'sdv_k_ap_prepare', f'profiles/{RND}_pmc_sdv_k_ap_prepare.json', 'apmc', 'tools/audio_prof.py 10000 1: 14.7 M sample pairs per launch, 176.4 MB read + 176.4 MB written algorithmic'], stdout=subprocess.DEVNULL)
# the PMC sets of tools/gpu_pmc_round3.sh (kernels the round-2 verdict asked counters for) and the visualiser's kernel stats
for kern, prefix, name, workload in (
        ('sdv_k_stitch_step', 'stpmc', 'sdv_k_stitch_step', 'tools/stitch_prof.py 10000 2 cont: 10 000-frame NTSC continuing tape, 4096 resident waves work 10 000 turns off a queue'),
        ('sdv_k_stitch_analyze', 'stpmc', 'sdv_k_stitch_analyze', 'tools/stitch_prof.py 10000 2 cont: 10 001 frames per launch'),
        ('sdv_k_pcm16_analyse', 'p16spmc', 'sdv_k_pcm16_analyse_si', 'tools/pcm16_prof.py 10000 1 si: 1024 frames per launch (one batch)'),
        ('sdv_k_pcm16_analyse', 'p16epmc', 'sdv_k_pcm16_analyse_ei', 'tools/pcm16_prof.py 10000 1 ei: 3072 frames per launch (three batches)'),
        ('sdv_k_pcm1_prescan', 'p1fpre', 'sdv_k_pcm1_prescan', 'tools/pcm1_frames_prof.py 10000 1: 10 000 frames, four prescan lines per frame, last mode of the run (NORMAL)'),
        ('sdv_k_pcm16_prescan', 'p16fpre', 'sdv_k_pcm16_prescan', 'tools/pcm16_frames_prof.py 10000 1: 10 000 frames, four prescan lines per frame, last mode of the run (NORMAL)'),
        ('sdv_k_pcm16_frames_bin', 'p16fpre', 'sdv_k_pcm16_frames_bin', 'tools/pcm16_frames_prof.py 10000 1: the frames the lean build gave up, last mode of the run (NORMAL)'),
        ('sdv_k_pcm16_frames_lean', 'p16fpre', 'sdv_k_pcm16_frames_lean', 'tools/pcm16_frames_prof.py 10000 1: 10 000 frames per launch, last mode of the run (NORMAL)'),
        ('sdv_k_pcm1_frames_lean', 'p1fpre', 'sdv_k_pcm1_frames_lean', 'tools/pcm1_frames_prof.py 10000 1: 10 000 frames per launch, last mode of the run (NORMAL)'),
        ('sdv_k_pcm1_frames_bin', 'p1fpre', 'sdv_k_pcm1_frames_bin', 'tools/pcm1_frames_prof.py 10000 1: the frames the lean build gave up, last mode of the run (NORMAL)'),
        ('sdv_k_ap_plan', 'applan', 'sdv_k_ap_plan', 'tools/audio_prof.py 10000 1: last tape of the run (an invalid word in every window)')):
    if glob.glob(f'gpurun_out/{prefix}1/**/*_counter_collection.csv', recursive=True):
        subprocess.check_call([sys.executable, 'tools/pmc_to_json.py', kern, f'profiles/{RND}_pmc_{name}.json', prefix, workload], stdout=subprocess.DEVNULL)
if glob.glob('gpurun_out/prof_vis/*/*kernel_stats.csv'):
    trim(newest('gpurun_out/prof_vis/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_vis_kernel_stats.csv')
d = json.loads(open('gpurun_out/bench_full.json').read().strip().split('\n')[-1])
if os.path.exists('gpurun_out/prof_audio.log'):
    shutil.copy('gpurun_out/prof_audio.log', f'profiles/{RND}_audio_prof.log')
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'launch ms', d['roofline']['avg_launch_ms'], 'traffic', d['roofline']['traffic'])
for k in ('stitch_stage', 'pcm1_stage', 'pcm1_front_stage', 'pcm16x0_stage', 'audio_stage'):
    if k in d: print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if a not in ('note', 'cpu_baseline')})
print('cpu', d['cpu_baseline']['value'], d['stitch_stage']['cpu_baseline']['value'], d['pcm1_stage']['cpu_baseline']['value'], d.get('host_fed', {}).get('h2d_gb_per_s'))

# round 4: the kernels of the damaged-tape path (tools/gpu_pmc_round4.sh), on the C3 PAL tape of tools/pal_trace.py
C3 = "tools/pal_trace.py 2000 both: 2000 PAL frames, every 97th line lost, a cell inverted on one line in 53"
JUMPS = "tools/jump_probe.py 10000 16: 10 000 NTSC frames, the data window jumps 16 times"
for kernel, prefix, workload in (("sdv_k_stc007_frames", "fullpmc", C3), ("sdv_k_stc007_frames_plain", "plainpmc", C3), ("sdv_k_stc007_frames_fat", "fatpmc", C3), ("sdv_k_stc007_sweep_levels", "swlpmc", C3), ("sdv_k_stc007_sweep_pick", "swppmc", C3), ("sdv_k_hist_carry", "hcpmc", JUMPS)):
    if glob.glob(f'gpurun_out/{prefix}1/**/*_counter_collection.csv', recursive=True):
        subprocess.check_call([sys.executable, 'tools/pmc_to_json.py', kernel, f'profiles/{RND}_pmc_{kernel}.json', prefix, workload], stdout=subprocess.DEVNULL)
if glob.glob('gpurun_out/prof_c3/*/*kernel_stats.csv'):
    trim(newest('gpurun_out/prof_c3/*/*kernel_stats.csv'), f'profiles/{RND}_rocprofv3_c3_tape_kernel_stats.csv')
