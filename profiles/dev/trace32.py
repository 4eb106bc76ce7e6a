"""decode the [pesto trace32] line of a PESTO_PROFILE_PHASES build (timeline of block 0 / wave 0 of the last traced launch)"""
import re
import sys
lines = [l for l in sys.stdin.read().split("\n") if "[pesto trace32]" in l]
names = {0: "gath_issue", 1: "geo+pr", 2: "L1(0,1)", 3: "keys+logits", 4: "softmax", 5: "pv+L1(2,3)", 6: "valL2", 7: "z3a", 8: "valL3", 9: "z3b+wsum",
         10: "finalize", 19: "ITEM", 20: "rows_setup", 37: "fin:issue", 38: "fin:barrier1", 39: "fin:rows+b2", 40: "fin:compute", 41: "fin:prepare", 31: "item_done"}
names.update({30: "p:setup", 32: "p:softmax", 33: "p2:L1+L2", 34: "p2:L3", 35: "p2:acc", 36: "p2:final", 50: "weights_in_lds", 51: "END"})
n = int(sys.argv[1]) if len(sys.argv) > 1 else 140
for l in lines:
    head, body = l.split("(last launch):")
    items = re.findall(r' (\d+):(-?\d+)', body)
    tot = sum(int(v) for _, v in items)
    print(head.replace("[pesto trace32] ", "").split(" id:")[0], f"total {tot} ticks:", " ".join(f"{names.get(int(k), k)}:{int(v)}" for k, v in items[:n]))
