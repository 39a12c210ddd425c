"""Static instruction histogram of one kernel by source line, from an assembly listing with line tables
(hipcc ... -gline-tables-only --cuda-device-only -S file.hip -o file.s).  A first look at where a kernel's vector
instructions come from; static counts (a loop body counts once).
usage: asm_lines.py file.s <mangled-kernel-name-substring> [top]"""
import sys, re, collections
path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
inside = False; cur = None
for ln in open(path, errors="replace"):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if not inside:
        if s.endswith(":") and key in s and not s.startswith("."):
            inside = True
        elif re.match(r'^[_A-Za-z0-9]+:', s) and key in s.split(":")[0]:
            inside = True
        continue
    if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
        break
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)(\w+)', s)
    if m and cur:
        unit = {"v_": "valu", "s_": "salu", "ds_": "lds"}.get(m.group(1), "mem")
        cnt[cur] += 1; kinds[cur][unit] += 1
tot = collections.Counter()
for k, c in kinds.items():
    for u, n in c.items(): tot[u] += n
print("total", dict(tot))
for (f, l), n in cnt.most_common(top):
    print(f"{f}:{l:5d} {n:5d} ", dict(kinds[(f, l)]))
