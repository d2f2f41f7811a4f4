#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/profile_fp16.sh (the FP16 entry: GoogLeNet as an FP16 IR, batch 256) into the summaries kept under profiles/:
profiles/<tag>_fp16_kernel_stats.csv (rocprofv3 --stats), profiles/<tag>_fp16_traffic.json (HBM bytes per launch of the f16 convolution kernels and
the other launches of the pass; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and profiles/<tag>_fp16_summary.md."""
import csv, glob, json, os, sys


_demangled = {}


def short(name):
    if name.startswith('_Z'):          # rocprofv3 leaves some names mangled
        if name not in _demangled:
            try:
                import subprocess
                _demangled[name] = subprocess.run(['c++filt', name], capture_output=True, text=True, check=True).stdout.strip() or name
            except Exception:
                _demangled[name] = name
        name = _demangled[name]
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def is_conv(name):
    return ('conv_f16' in name or 'conv_igemm' in name or 'conv_pool1x1' in name) and 'pack_kernel' not in name


def family(name):
    return short(name).split('<')[0]


def find(root, pattern):
    hits = glob.glob(os.path.join(root, '**', pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def main():
    raw, tag = sys.argv[1], sys.argv[2]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(repo, 'profiles')
    md = ['# rocprofv3 summary of the FP16 entry `{}`'.format(tag), '',
          'Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --fp16-passes 8` (googlenet-v1 as an FP16 IR read with fp16_as_fp32=False, batch 256, '
          'eager passes on one stream: a launch\'s duration is its own).', '']
    stats = find(os.path.join(raw, 'stats'), '*kernel_stats.csv')
    if stats:
        rows = list(csv.DictReader(open(stats)))
        open(os.path.join(prof, tag + '_fp16_kernel_stats.csv'), 'w').write(open(stats).read())
        md += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
        tot = calls = 0.0
        for r in rows[:30]:
            md.append('| `{}` | {} | {:.3f} | {:.2f} | {} |'.format(short(r['Name'])[:80], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage']))
        for r in rows:
            if is_conv(r['Name']):
                tot += float(r['TotalDurationNs']); calls += int(r['Calls'])
        if calls:
            md += ['', '**f16 convolution kernels, all instantiations:** {:.0f} launches, {:.3f} ms total, average {:.2f} us per launch.'.format(calls, tot / 1e6, tot / calls / 1e3)]
    trace = find(os.path.join(raw, 'stats'), '*kernel_trace.csv')
    if trace:
        # the last 5 passes: per-pass device time of the convolution launches and of everything
        rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(trace))), key=lambda t: t[0])
        conv = [(b, e) for b, e, n in rows if is_conv(n)]
        n_stem = sum(1 for _, _, n in rows if 'conv_f16_stem_kernel' in n)
        if n_stem and conv:
            per_pass = len(conv) // n_stem
            tail = conv[-5 * per_pass:]
            md += ['', 'Last five passes: {} convolution launches per pass, **{:.3f} ms** of convolution launches per pass ({:.2f} us per launch).'.format(
                per_pass, sum(e - b for b, e in tail) / 5e6, sum(e - b for b, e in tail) / len(tail) / 1e3)]
    traffic = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        cc = find(os.path.join(raw, 'pmc_' + counter), '*counter_collection.csv')
        if not cc:
            continue
        agg = {}
        for r in csv.DictReader(open(cc)):
            if r['Counter_Name'] != counter:
                continue
            for fam in (['convolution_kernels', 'conv:' + family(r['Kernel_Name'])] if is_conv(r['Kernel_Name']) else [family(r['Kernel_Name'])]):
                a = agg.setdefault(fam, [0.0, 0]); a[0] += float(r['Counter_Value']); a[1] += 1
        traffic[counter] = agg
    if traffic.get('FETCH_SIZE') and traffic.get('WRITE_SIZE'):
        out = {'tag': tag, 'tree': os.environ.get('PVHIP_TREE', 'unknown'), 'note': 'KB counters from separate --pmc passes of `bench.py --fp16-passes 3`; read side doubled '
               '(gfx950 FETCH_SIZE reports half of a wide coalesced stream, MI355X_MICROARCH.md section HBM); bytes per launch', 'kernels': {}, 'convolution_kernels_by_family': {}}
        md += ['', '## HBM traffic per launch (PMC, corrected)', '', '| kernel family | launches | read MB | write MB | total MB |', '|---|---|---|---|---|']
        for fam, (kb, n) in sorted(traffic['FETCH_SIZE'].items(), key=lambda kv: -kv[1][0]):
            w = traffic['WRITE_SIZE'].get(fam)
            if not w or not n:
                continue
            rd, wr = 2.0 * kb * 1024.0 / n, w[0] * 1024.0 / max(1, w[1])
            rec = {'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'launches_sampled': n}
            if fam.startswith('conv:'):
                out['convolution_kernels_by_family'][fam[5:]] = rec
            else:
                out['kernels'][fam] = rec
            if rd + wr > 1e5:
                md.append('| `{}` | {} | {:.2f} | {:.2f} | {:.2f} |'.format(fam, n, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
        json.dump(out, open(os.path.join(prof, tag + '_fp16_traffic.json'), 'w'), indent=1)
    sq = find(os.path.join(raw, 'pmc_SQ'), '*counter_collection.csv')
    if sq:
        fam_agg = {}
        for r in csv.DictReader(open(sq)):
            if is_conv(r['Kernel_Name']):
                fa = fam_agg.setdefault(family(r['Kernel_Name']), {})
                fa[r['Counter_Name']] = fa.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        tot_c = sum(fa.get('GRBM_GUI_ACTIVE', 0.0) for fa in fam_agg.values()) / 8.0
        if tot_c:
            md += ['', '## f16 convolution kernels, SQ counters summed over their launches', '',
                   '| kernel | MFMA pipe busy, % of SIMD-cycles | CUs holding a wave, % | share of the convolution cycles, % |', '|---|---|---|---|']
            for fam, fa in sorted(fam_agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0.0)):
                c = fa.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
                if c:
                    md.append('| `{}` | {:.1f} | {:.1f} | {:.1f} |'.format(fam, 100.0 * fa.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / c,
                                                                     100.0 * fa.get('SQ_BUSY_CU_CYCLES', 0.0) / 256.0 / c, 100.0 * c / tot_c))
            busy = sum(fa.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for fa in fam_agg.values())
            md += ['', 'All f16 convolution launches: MFMA pipe busy {:.1f} % of SIMD-cycles.'.format(100.0 * busy / 1024.0 / tot_c)]
    open(os.path.join(prof, tag + '_fp16_summary.md'), 'w').write('\n'.join(md) + '\n')
    print('\n'.join(md))


if __name__ == '__main__':
    main()
