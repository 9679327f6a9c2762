"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid) and print total / calls / average duration."""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:90]


def main(path, top=30):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Grid_Size_Y'], r['Grid_Size_Z'])
        agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    rows = sorted(((sum(v), len(v), sum(v) / len(v), k) for k, v in agg.items()), reverse=True)[:top]
    print('# total_us   calls     avg_us   kernel [blocks_x, grid_y, grid_z]')
    for tot, n, avg, k in rows:
        print('%12.1f %6d %12.1f   %s [%d,%s,%s]' % (tot, n, avg, k[0], k[1], k[2], k[3]))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
