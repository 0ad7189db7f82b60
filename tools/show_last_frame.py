import sys
for fn in sys.argv[1:]:
    L=[l.rstrip() for l in open(fn)]
    idx=max(i for i,l in enumerate(L) if "k_primary" in l)
    print("==",fn)
    t0=float(L[idx].split()[0])
    for l in L[idx:]:
        p=l.split(None,5)
        print(f"{float(p[0])-t0:9.1f} +{float(p[3]):8.1f}  {p[5][:60]}")
