"""Build-time audit for a hipcc 7.2 miscompile met in round 2: a select on a value loaded from global memory was emitted
as v_cmp (VCC) followed by s_cselect (SCC), i.e. selecting on the carry of an unrelated s_addc. Scans device assembly for an
SCC consumer whose last SCC writer is an address add / shift. Usage: scan_scc.py file.s [...]; exit code 1 on a hit."""
import re, sys
bad = 0
for path in sys.argv[1:]:
    cur, last = None, None
    for i, l in enumerate(open(path)):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur, last = m.group(1), None
            continue
        t = l.strip().split(' ')[0] if l.strip() else ''
        if not t or t[0] in ';.':
            continue
        if (t.startswith('s_cselect') or t.startswith('s_cbranch_scc')) and last and re.match(r's_(add|sub|lshl|ashr|lshr)', last):
            print("SUSPECT %s:%d %s: %s after %s" % (path, i + 1, cur, t, last))
            bad = 1
        if re.match(r's_(add|sub|addc|subb|and|or|xor|andn2|orn2|nand|nor|xnor|not|lshl|lshr|ashr|bfe|cmp|bitcmp|min|max|abs|absdiff|wqm|bcnt|ff|flbit|sext)', t):
            last = t
sys.exit(bad)
