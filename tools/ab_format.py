import re,collections,sys
rows=collections.OrderedDict(); cur=None
for l in open(sys.argv[1]):
    if l.startswith("== "): cur=l.split()[1]; continue
    m=re.match(r"(.{38})\s*([0-9.]+) us",l)
    if m: rows.setdefault(m.group(1).strip(),{}).setdefault(cur,[]).append(float(m.group(2)))
for k,v in rows.items():
    a=sum(v["ab"])/len(v["ab"]); p=sum(v["prod"])/len(v["prod"])
    print("%-42s ab %7.1f  prod %7.1f  %+5.1f%%"%(k,a,p,(p/a-1)*100))
