"""Timeline of one cfg2 step (N=1e5, M=512): `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/cfg2_trace.py` then
`python tools/cfg2_trace.py DIR` lists every kernel of the LAST step (start / end / duration, queue) and the idle gaps of the device."""
import sys, csv, glob, os
if len(sys.argv) > 1 and os.path.isdir(sys.argv[1]):
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    def short(n):
        n = n.split('(')[0]
        return n[n.find('gemm_f64_kernel'):][:64] if 'gemm_f64_kernel' in n else n[-44:]
    # steps are separated by the largest idle gaps (host work between calls)
    ends = [int(r['End_Timestamp']) for r in rows]
    starts = [int(r['Start_Timestamp']) for r in rows]
    gaps = sorted(((starts[i + 1] - max(ends[:i + 1][-64:]), i) for i in range(len(rows) - 1)), reverse=True)
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cuts = sorted(i for _, i in gaps[:nsteps - 1])
    last = rows[cuts[-1] + 1:]
    t0 = int(last[0]['Start_Timestamp'])
    busy_end = t0; idle = 0.0
    for r in last:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > busy_end: idle += (s - busy_end) / 1e3
        busy_end = max(busy_end, e)
        print('%9.1f %9.1f  %7.1f us  q%s  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), short(r['Kernel_Name'])))
    print('step span %.1f us, %d kernels, device idle %.1f us' % ((busy_end - t0) / 1e3, len(last), idle))
else:
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zero-inflated-gp_amd'))
    import bench, zigp, torch
    X, Y, p = bench.synth(100000, 512, 3)
    e = zigp.DenseEngine(0)
    e.set_data_device(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
    for _ in range(4):
        e.elbo(p)
        torch.cuda.synchronize()
