#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/profile_bench.sh into the small summaries kept under profiles/:

  profiles/<tag>_kernel_stats.csv   per-kernel calls / total / average / percentage (rocprofv3 --stats)
  profiles/<tag>_summary.md         the same, readable, plus the conv_igemm aggregate the bench's roofline uses
  profiles/<tag>_traffic.json       HBM bytes per launch of the dominant kernel family from the PMC passes
                                    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, both in KB)
"""
import csv
import glob
import json
import os
import sys


def is_conv(name):
    """The kernels behind the Convolution nodes: implicit GEMM (also with the MaxPool in front folded in), the Winograd forms, the pointwise kernel."""
    return 'conv_igemm' in name or 'conv_wino' in name or 'conv_pool1x1' in name or 'conv_pw_kernel' in name or 'conv_stem_f32_kernel' in name or 'conv_stem_wino_kernel' in name


def conv_family(name):
    """Which convolution kernel: the split of `convolution_kernels` (profiles/<tag>_traffic.json: convolution_kernels_by_family)."""
    for key, fam in (('conv_wino4s_kernel', 'conv_wino4s_kernel'), ('conv_wino4_kernel', 'conv_wino4_kernel'), ('conv_wino_kernel', 'conv_wino_kernel'), ('conv_pw_kernel', 'conv_pw_kernel'),
                     ('conv_pool1x1_kernel', 'conv_pool1x1_kernel'), ('conv_stem_f32_kernel', 'conv_stem_kernel'), ('conv_stem_wino_kernel', 'conv_stem_kernel'), ('conv_igemm_dma_kernel', 'conv_igemm_dma_kernel'), ('conv_igemm', 'conv_igemm_other')):
        if key in name:
            return fam
    return None


def find(root, pattern):
    hits = glob.glob(os.path.join(root, '**', pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None      # gpurun MERGES into gpurun_out/: an older run's files may still be there


def main():
    raw, tag = sys.argv[1], sys.argv[2]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(repo, 'profiles')
    os.makedirs(prof, exist_ok=True)
    md = ['# rocprofv3 summary `{}`'.format(tag), '',
          'Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-images 0 --no-extra --min-seconds 0 --requests 1 --streams 1` '
          '(googlenet-v1, batch 256, 1 GPU; one request at a time on ONE compute stream, so that a launch\'s start-to-end time is its own: '
          'this is what bench.py\'s roofline measures on its sampled steps).  The default command (several whole-batch requests in flight, '
          'kernels of different passes overlapping) is summarised at the end.', '']
    stats = find(os.path.join(raw, 'stats'), '*kernel_stats.csv')
    conv_total_ns = conv_calls = 0
    per_pass = 57                      # convolution launches per forward pass (bench.py says how many after fusion)
    line = os.path.join(raw, 'bench_line_under_profiler.json')
    if os.path.isfile(line) and os.path.getsize(line):
        per_pass = int(json.loads(open(line).read())['roofline'].get('launches_per_step', per_pass))
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(prof, tag + '_kernel_stats.csv'), 'w') as f:
            f.write(open(stats).read())
        md += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
        for r in rows:
            name = r['Name']
            short = name.replace('(anonymous namespace)::', '').split('(')[0][:70]
            md.append('| `{}` | {} | {:.3f} | {:.2f} | {} |'.format(short, r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                                  float(r['AverageNs']) / 1e3, r['Percentage']))
            if is_conv(name):
                conv_total_ns += float(r['TotalDurationNs'])
                conv_calls += int(r['Calls'])
        if conv_calls:
            md += ['', '**conv_wino4s_kernel + conv_wino4_kernel + conv_wino_kernel + conv_pw_kernel + conv_stem_f32_kernel + conv_igemm_dma_kernel + conv_pool1x1_kernel, all instantiations:** {} launches, {:.3f} ms total, **average {:.2f} us per launch** '
                   '({:.3f} ms per forward pass of {} launches).'.format(conv_calls, conv_total_ns / 1e6, conv_total_ns / conv_calls / 1e3,
                                                                         conv_total_ns / conv_calls * per_pass / 1e6, per_pass)]
    trace = find(os.path.join(raw, 'stats'), '*kernel_trace.csv')
    if trace and conv_calls:
        conv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(trace)) if is_conv(r['Kernel_Name']))
        tail_ = conv[-5 * per_pass:]
        head_ = conv[:3 * per_pass]
        md += ['', 'From the kernel trace of the same run: the first 3 forward passes (clocks ramping up from idle, cold caches) average {:.2f} us '
               'per convolution launch, the LAST 5 passes **{:.2f} us** -- the steady state the timed region of bench.py sees.'.format(
                   sum(e - b for b, e in head_) / len(head_) / 1e3, sum(e - b for b, e in tail_) / len(tail_) / 1e3)]
    line = os.path.join(raw, 'bench_line_under_profiler.json')
    if os.path.isfile(line) and os.path.getsize(line):
        b = json.loads(open(line).read())
        r = b['roofline']
        md += ['', 'bench.py line of the profiled run: {:.1f} images/s, {:.3f} ms/step; roofline.achieved {:.2f} TFLOP/s = '
               '{:.3f} GFLOP per launch / {:.2f} us average launch (hipEvents on the compute stream, {} sampled steps).'.format(
                   b['value'], b['ms_per_step'], r['achieved'], r['flops_per_launch'] / 1e9, r['avg_launch_us'], r.get('event_sampled_steps'))]
    forked = find(os.path.join(raw, 'stats_forked'), '*kernel_stats.csv')
    forked_md = []
    if forked:
        tot = calls = 0.0
        for r in csv.DictReader(open(forked)):
            if is_conv(r['Name']):
                tot += float(r['TotalDurationNs'])
                calls += int(r['Calls'])
        with open(os.path.join(prof, tag + '_kernel_stats_forked.csv'), 'w') as f:
            f.write(open(forked).read())
        forked_md = ['', '## Default command (requests in flight)', '',
                     '`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-images 0 --no-extra --min-seconds 0`: per-kernel table in `{}_kernel_stats_forked.csv`.  '
                     'Convolution launches: {:.0f}, average start-to-end {:.2f} us -- longer than alone on one stream because launches of different '
                     'passes share the chip; the step is shorter.'.format(tag, calls, tot / max(1.0, calls) / 1e3)]
        fl = os.path.join(raw, 'bench_line_forked.json')
        if os.path.isfile(fl) and os.path.getsize(fl):
            b = json.loads(open(fl).read())
            forked_md.append('bench.py line of that run: {:.1f} images/s, {:.3f} ms/step.'.format(b['value'], b['ms_per_step']))
    traffic = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        cc = find(os.path.join(raw, 'pmc_' + counter), '*counter_collection.csv')
        if not cc:
            continue
        per_kernel = {}
        for r in csv.DictReader(open(cc)):
            if r['Counter_Name'] != counter:
                continue
            fam = 'convolution_kernels' if (is_conv(r['Kernel_Name'])) else r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('<')[0].replace('void ', '')
            agg = per_kernel.setdefault(fam, [0.0, 0])
            agg[0] += float(r['Counter_Value'])
            agg[1] += 1
            if fam == 'convolution_kernels':       # ... and split by kernel
                agg = per_kernel.setdefault('conv:' + conv_family(r['Kernel_Name']), [0.0, 0])
                agg[0] += float(r['Counter_Value'])
                agg[1] += 1
        traffic[counter] = {k: {'sum_kb': v[0], 'launches': v[1]} for k, v in per_kernel.items()}
    if traffic.get('FETCH_SIZE') and traffic.get('WRITE_SIZE'):
        out = {'tag': tag, 'tree': os.environ.get('PVHIP_TREE', 'unknown'), 'note': 'KB counters from separate --pmc passes; read side doubled (gfx950 FETCH_SIZE reports half of a wide '
                                   'coalesced stream, MI355X_MICROARCH.md section HBM); bytes per launch', 'kernels': {}}
        md += ['', '## HBM traffic per launch (PMC, corrected)', '', '| kernel family | launches | read MB | write MB | total MB |', '|---|---|---|---|---|']
        for fam, f in traffic['FETCH_SIZE'].items():
            w = traffic['WRITE_SIZE'].get(fam)
            if not w or not f['launches']:
                continue
            rd = 2.0 * f['sum_kb'] * 1024.0 / f['launches']
            wr = w['sum_kb'] * 1024.0 / max(1, w['launches'])
            if fam.startswith('conv:'):
                passes = max(1.0, traffic['FETCH_SIZE']['convolution_kernels']['launches'] / float(per_pass))
                out.setdefault('convolution_kernels_by_family', {})[fam[5:]] = {
                    'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'launches_sampled': f['launches'], 'launches_per_step': round(f['launches'] / passes, 2)}
            else:
                out['kernels'][fam] = {'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'launches_sampled': f['launches']}
            md.append('| `{}` | {} | {:.2f} | {:.2f} | {:.2f} |'.format(fam, f['launches'], rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
        with open(os.path.join(prof, tag + '_traffic.json'), 'w') as fo:
            json.dump(out, fo, indent=1)
    sq = find(os.path.join(raw, 'pmc_SQ'), '*counter_collection.csv')
    if sq:
        agg, fam_agg = {}, {}
        for r in csv.DictReader(open(sq)):
            if is_conv(r['Kernel_Name']):
                agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
                fa = fam_agg.setdefault(conv_family(r['Kernel_Name']), {})
                fa[r['Counter_Name']] = fa.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if agg.get('GRBM_GUI_ACTIVE'):
            cyc = agg['GRBM_GUI_ACTIVE'] / 8.0
            md += ['', '## convolution kernels (conv_wino4s + conv_wino4 + conv_wino + conv_pw + conv_igemm_dma + conv_pool1x1), SQ counters summed over their launches', '']
            md += ['- MFMA pipe busy: {:.1f} % of SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8))'.format(
                100.0 * agg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / cyc)]
            if agg.get('SQ_BUSY_CU_CYCLES'):
                md += ['- CUs holding at least one wave: {:.1f} % of CU-cycles (SQ_BUSY_CU_CYCLES / 256 CUs / (GRBM_GUI_ACTIVE / 8))'.format(
                    100.0 * agg['SQ_BUSY_CU_CYCLES'] / 256.0 / cyc)]
            kt = find(os.path.join(raw, 'pmc_SQ'), '*kernel_trace.csv')
            if kt:
                dur = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt)) if is_conv(r['Kernel_Name']))
                if dur:
                    ghz = cyc / dur
                    md += ['- GRBM_GUI_ACTIVE / 8 / summed kernel time = {:.2f} GHz: an UPPER bound of the shader clock (the quotient reads high on '
                           'dispatches shorter than ~0.3 ms, and these are 0.03-0.7 ms); s_memtime / s_memrealtime stamps inside the Winograd kernels '
                           '(diagnostic build, scripts/stamps_wino4.py) read 1.98-2.10 GHz, and fp32 MFMA alone sustains 141-148 TFLOP/s at 2.26-2.40 GHz '
                           '(bench.py: roofline.sustained)'.format(ghz)]
            for k in sorted(agg):
                md.append('- {} = {:.4g}'.format(k, agg[k]))
            md += ['', '| kernel | MFMA pipe busy, % of SIMD-cycles | CUs holding a wave, % | share of the convolution cycles, % |', '|---|---|---|---|']
            for fam, fa in sorted(fam_agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0.0)):
                c = fa.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
                if c:
                    md.append('| `{}` | {:.1f} | {:.1f} | {:.1f} |'.format(fam, 100.0 * fa.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / c,
                                                                     100.0 * fa.get('SQ_BUSY_CU_CYCLES', 0.0) / 256.0 / c, 100.0 * c / cyc))
    md += forked_md
    with open(os.path.join(prof, tag + '_summary.md'), 'w') as f:
        f.write('\n'.join(md) + '\n')
    print('\n'.join(md))


if __name__ == '__main__':
    main()
