import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
e = max(float((a["emb"][k] - b["emb"][k]).abs().max()) for k in a["emb"])
errs = sorted(((float((a["g"][n] - b["g"][n]).abs().max() / (a["g"][n].abs().max() + 1e-12)), n) for n in a["g"]), reverse=True)
print(sys.argv[1], sys.argv[2], "emb diff", e, "params over 5e-3:", len([x for x in errs if x[0] > 5e-3]), errs[:3])
