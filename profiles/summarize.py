"""Turn a rocprofv3 rocpd sqlite result (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`) into
the per-kernel summary committed under profiles/.   python profiles/summarize.py DB [OUT.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
        "max(grid_x), max(grid_y), max(workgroup_x), max(lds_size), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(scratch_size) "
        # (one row per kernel AND grid: the split scan's audit runs the same kernels once on the un-split batch -- 25 instead of
        # 125 work-groups per direction, five times the columns each -- and must not be averaged into the timed launches)
        "from kernels group by name, grid_x, grid_y order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    lines = ["kernel,calls,avg_us,min_us,max_us,total_ms,percent,grid_x,grid_y,wg_x,lds_bytes,vgpr,agpr,sgpr,scratch"]
    for r in rows:
        lines.append(",".join([f'"{r[0]}"', str(r[1]), f"{r[2]/1e3:.1f}", f"{r[3]/1e3:.1f}",
                               f"{r[4]/1e3:.1f}", f"{r[5]/1e6:.3f}", f"{100*r[5]/tot:.2f}"] +
                              [str(v) for v in r[6:]]))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
