"""Joins a rocprofv3 --kernel-trace CSV of bench.py with the library's own launch log ($ALGP_LAUNCH_LOG: class m n k ... per
GEMM launch, in enqueue order) by dispatch order and prints, for the candidate solve's GEMM launches (class 2), a table by
launch shape (tiles x K): calls, summed duration, TFLOP/s executed -- and the sum of their durations per solve, the figure
bench.py's roofline.serial_kernel_frac is formed from with HIP events.
Also printed, per solve, the SPAN of its launches -- first dispatch start to last dispatch end in the trace -- beside the sum of
their durations: with the solve's row chunks on three streams the launches overlap, the span is the solve's wall time as the
profiler saw it, the figure bench.py's roofline.achieved is formed from with HIP events (wall_ms_all_launches / steps).
usage: python tools/trace_shapes.py <kernel_trace.csv> <launch_log.txt> <solves in the trace> [<bench line .json>]"""
import csv
import json
import sys
from collections import defaultdict

trace, log, nsolves = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(trace)) if 'gemm_nt_kernel_dma4' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Dispatch_Id']))
launches = [tuple(int(v) for v in ln.split()) for ln in open(log) if ln.strip()]
if len(rows) != len(launches):
    raise SystemExit('trace holds %d GEMM dispatches, the launch log %d lines' % (len(rows), len(launches)))
shape = defaultdict(lambda: [0, 0.0, 0.0])
tot_ns, tot_fl, n2 = 0, 0.0, 0
solve_rows = []
for r, rec in zip(rows, launches):
    klass, m, n, k, lower, batch, ktri, es = rec[:8]
    kcut = rec[8] if len(rec) > 8 else 0
    tiles = int(r['Grid_Size_X']) // 256
    want = (m // 128) * (m // 128 + 1) // 2 if lower else (m // 128) * (n // 128)
    if not ktri and tiles != want:
        raise SystemExit('dispatch %s: grid of %d tiles, the log says %d x %d (%d tiles)' % (r['Dispatch_Id'], tiles, m, n, want))
    if klass != 2:
        continue
    ns = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    fl = 2.0 * 128 * 128 * k * tiles * batch
    if kcut:                                   # column tile c of n / 128 walks k < 128 (c + 1) only (a block's explicit inverse)
        fl *= (n // 128 + 1) / (2.0 * (n // 128))
    s = shape[(tiles, -k if kcut else k)]
    s[0] += 1
    s[1] += ns * 1e-6
    s[2] += fl
    tot_ns += ns
    tot_fl += fl
    n2 += 1
    solve_rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), ns))
print('candidate-solve GEMM launches (class GEMM_TRSM): %d in %d solves' % (n2, nsolves))
print('%10s %7s %7s %11s %10s' % ('tiles', 'K', 'calls', 'sum ms', 'TFLOP/s'))
for (tiles, k), (calls, ms, fl) in sorted(shape.items(), key=lambda kv: -kv[1][1]):
    print('%10d %7s %7d %11.3f %10.1f' % (tiles, ('<=%d' % -k) if k < 0 else str(k), calls, ms, fl / (ms * 1e-3) / 1e12))
per = tot_ns * 1e-6 / nsolves
print('sum of launch durations per solve: %.2f ms (%d launches per solve); executed flop per solve %.4g -> %.1f TFLOP/s executed'
      % (per, n2 // nsolves, tot_fl / nsolves, tot_fl / nsolves / (per * 1e-3) / 1e12))
if len(sys.argv) > 4:
    b = json.load(open(sys.argv[4]))
    rf = b['roofline']
    K = b['steps']
    ev = rf['sum_launch_ms'] / K
    print('bench line of the same process: HIP-event sum of the launches %.2f ms per solve (trace / events = %.3f); algorithmic N^2 M = %.4g flop'
          % (ev, per / ev, rf['algorithmic_flops_per_step']))
    print('  -> the launches back to back: %.1f TFLOP/s algorithmic = %.3f of %.1f by the trace, %.3f by the events'
          % (rf['algorithmic_flops_per_step'] / (per * 1e-3) / 1e12, rf['algorithmic_flops_per_step'] / (per * 1e-3) / 1e12 / rf['peak'], rf['peak'],
             rf['algorithmic_flops_per_step'] / (ev * 1e-3) / 1e12 / rf['peak']))

# per solve: span (first start -> last end) beside the sum of the launch durations
if n2 % nsolves == 0 and n2 > 0:
    per_solve = n2 // nsolves
    spans, sums, streams = [], [], set(r['Stream_Id'] for r in rows) if rows and 'Stream_Id' in rows[0] else set()
    # round 6: a solve whose train set ends in a narrow last tile finishes with the tail kernel (tail.hip) -- part of the solve's
    # span in the bench's events, so also here: the tail launches of the trace, dealt to the solves in dispatch order
    tails = sorted((r for r in csv.DictReader(open(trace)) if 'tail_cols_kernel' in r['Kernel_Name'] or 'tail_part_kernel' in r['Kernel_Name']
                    or 'tail_finish_kernel' in r['Kernel_Name']), key=lambda r: int(r['Dispatch_Id']))
    per_tail = len(tails) // nsolves if tails and len(tails) % nsolves == 0 else 0
    for q in range(nsolves):
        grp = solve_rows[q * per_solve:(q + 1) * per_solve]
        end = max(e for _, e, _ in grp)
        tail_ns = 0
        for r in tails[q * per_tail:(q + 1) * per_tail]:
            end = max(end, int(r['End_Timestamp']))
            tail_ns += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        spans.append((end - min(st for st, _, _ in grp)) * 1e-6)
        sums.append((sum(ns for _, _, ns in grp) + tail_ns) * 1e-6)
    if per_tail:
        print('(each solve ends with %d tail-kernel launch(es) for the narrow last tile: inside the spans below)' % per_tail)
    print('per solve, span of its launches (first start -> last end) / sum of their durations, ms:')
    print('  ' + '  '.join('%.2f / %.2f' % (a, b) for a, b in zip(spans, sums)))
    med = sorted(spans)[len(spans) // 2]
    print('median span %.2f ms over %d solves (%d GEMM streams in the trace); sum / span = %.2f'
          % (med, nsolves, len(streams), sorted(sums)[len(sums) // 2] / med))
    if len(sys.argv) > 4:
        b = json.load(open(sys.argv[4]))
        rf = b['roofline']
        wall = rf['wall_ms_all_launches'] / b['steps']
        # the bench's K timed steps are solves 1 + warmup .. warmup + K of the process (its stage timers bracket exactly them)
        w0 = b.get('warmup', 0)
        timed = spans[w0:w0 + b['steps']] if len(spans) >= w0 + b['steps'] else spans
        tm = sum(timed) / len(timed)
        print('bench line of the same process: wall time of the solve by HIP events %.2f ms (roofline.wall_ms_all_launches / steps); '
              'mean span of the same %d solves in the trace %.2f ms -> trace / events = %.3f'
              % (wall, len(timed), tm, tm / wall))
        print('  -> N^2 M / span = %.1f TFLOP/s = %.3f of %.1f by the trace; the line says %.3f'
              % (rf['algorithmic_flops_per_step'] / (tm * 1e-3) / 1e12, rf['algorithmic_flops_per_step'] / (tm * 1e-3) / 1e12 / rf['peak'],
                 rf['peak'], rf['frac']))
