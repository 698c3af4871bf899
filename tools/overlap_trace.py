"""Timeline of one overlapped cfg3 chunk (zigp_set_overlap(1)): run `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/overlap_trace.py`
then `python tools/overlap_trace.py DIR` prints, for a chunk in the middle of the last step, every kernel's start / end relative to the
chunk's first GEMM and which kernels ran at the same time."""
import sys, csv, glob, os
if len(sys.argv) > 1:
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    def short(n):
        n = n.split('(')[0]
        return n[n.find('gemm_f64_kernel'):][:60] if 'gemm_f64_kernel' in n else n[-40:]
    # the A1 launches (EpiStoreColsum with TRI 1) mark chunk starts; take the 20th-from-last pair (f, g)
    a1 = [i for i, r in enumerate(rows) if 'EpiStoreColsum' in r['Kernel_Name'] and ', 1, 8,' in r['Kernel_Name']]
    i0, i1 = a1[-40], a1[-38]
    t0 = int(rows[i0]['Start_Timestamp'])
    for r in rows[i0:i1]:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        print('%9.1f %9.1f  %7.1f us  q%s  %s' % (s, e, e - s, r.get('Queue_Id', '?'), short(r['Kernel_Name'])))
    print('chunk span %.1f us' % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3))
else:
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
    import bench, zigp, torch
    X, Y, p = bench.synth(1000000, 1024, 3)
    e = zigp.DenseEngine(0)
    e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    e.set_overlap(1)
    for _ in range(2):
        e.elbo(p)
