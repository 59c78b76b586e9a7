import subprocess, time, re, sys
for args in (["dirty", "0"], ["dirty", "4"], ["dirty", "16"], ["dirty", "32"], ["dirty", "64"], ["dirty", "32", "free"], ["dirty", "64", "free"], ["dirty", "16"], ["dirty", "0"]):
    r = subprocess.run(["./tools/micro/devalloc"] + args, stdout=subprocess.PIPE, universal_newlines=True)
    back = time.time()
    m = re.search(r"leaving at ([0-9.]+)", r.stdout)
    f = re.search(r"hipFree of all of it ([0-9.]+) ms", r.stdout)
    print("%-20s from _exit until the parent has it back: %.3f s%s" % (" ".join(args), back - float(m.group(1)), ("   (hipFree before: %s ms)" % f.group(1)) if f else ""), flush=True)
