"""decode the [pesto trace32] line of a PESTO_PROFILE_PHASES build (timeline of block 0 / wave 0 of the last traced launch)"""
import re
import sys
s = sys.stdin.read()
items = re.findall(r' (\d+):(-?\d+)', s)
names = {0: "gath_issue", 1: "geo+pr", 2: "L1(0,1)", 3: "keys+logits", 4: "softmax", 5: "pv+L1(2,3)", 6: "valL2", 7: "z3a", 8: "valL3", 9: "z3b+wsum",
         10: "finalize", 19: "ITEM", 20: "rows_setup", 37: "fin:issue", 38: "fin:barrier1", 39: "fin:rows+b2", 40: "fin:compute", 41: "fin:prepare", 31: "item_done"}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 140
print(" ".join(f"{names.get(int(k), k)}:{int(v)}" for k, v in items[:n]))
