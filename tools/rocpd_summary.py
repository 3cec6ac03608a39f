"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into a small markdown table for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/r1_c2_kernel_stats.md "command line"

Kernels are grouped by (name, grid size) so that one kernel used at two problem sizes (e.g. the flash kernel for
self- and cross-attention) is reported separately — the self-attention row is the one bench.py's roofline quotes.
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name if len(name) < 70 else name[:67] + "..."


def main(db_path, out_path, cmd=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    sel = f"name, {gx}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start)" if gx else \
        "name, 0, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start)"
    rows = list(cur.execute(f"select {sel} from kernels group by name{', ' + gx if gx else ''} order by 4 desc"))
    total = sum(r[3] for r in rows) or 1
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        f.write("| kernel | grid.x (threads) | calls | total ms | avg us | min us | max us | % GPU time |\n|---|---|---|---|---|---|---|---|\n")
        for n, g, c, tot, avg, mn, mx in rows[:24]:
            f.write(f"| `{short(n)}` | {g} | {c} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.2f} |\n")
        f.write(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[2] for r in rows)} dispatches\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
