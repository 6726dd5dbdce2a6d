import sys, torch
res = torch.load(sys.argv[1])
plain = res["plain"]
for k, v in res.items():
    if k == "plain": continue
    d = (plain["p32"] - v["p32"]).abs()
    print(k, "max |dp32|", float(d.max()), "frac != ", float((d > 0).float().mean()), "losses", v["losses"], plain["losses"], "norms", v["norms"], plain["norms"])
    idx = d.argmax().item()
    print("   argmax index", idx, "of", d.numel())
