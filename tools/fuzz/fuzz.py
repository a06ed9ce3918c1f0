"""Fuzzes the HOST scene layer (scene parser, OBJ reader, PNG / .hdr decoders, BVH build) under AddressSanitizer + UBSan.
CPU only (GPU sanitizers are not available on this pool).  Usage:  python tools/fuzz/fuzz.py [seed] [cases per target]
Builds tools/fuzz/harness.cpp with g++ -fsanitize=address,undefined into a temp dir, then feeds it truncated / bit-flipped /
spliced copies of the shipped scenes, of a generated textured scene's images, and of an OBJ file.  A case is BAD when the
harness reports a sanitizer error, exits non-zero or runs for more than two minutes; rejections with a message are fine.
The findings of round 1 (image headers promising gigabytes, RES 800000 800000) are regression-tested in
tests/test_textures_host.py::test_corrupt_headers_are_rejected_not_allocated."""
import os, random, subprocess, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
TMP = tempfile.mkdtemp(prefix="ptk_fuzz_")
CS = os.path.join(ROOT, "project3-cuda-path-tracer-1_amd", "csrc")
H = os.path.join(TMP, "harness")
subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + CS,
                       os.path.join(ROOT, "tools", "fuzz", "harness.cpp"), os.path.join(CS, "scene.cpp"), os.path.join(CS, "image.cpp"), "-lz", "-o", H])
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 3)
N=int(sys.argv[2]) if len(sys.argv)>2 else 60
W=os.path.join(TMP,'work'); os.makedirs(W+'/tex')
import texture_scene
scene_path = texture_scene.make(W+'/tex')
def run(path, cwd=None):
    r = subprocess.run([H, path], capture_output=True, text=True, errors="replace", cwd=cwd, timeout=120)
    bad = ('ERROR: AddressSanitizer' in r.stderr) or ('runtime error' in r.stderr) or r.returncode not in (0,)
    return bad, r
bad, r = run(str(scene_path)); print('baseline textured', bad, r.stdout.strip()[-100:], r.stderr[-200:])
sp=str(scene_path)
targets=[ROOT+'/scenes/cornell.txt',ROOT+'/scenes/bunny_cornell.txt',ROOT+'/scenes/obj_test.txt', sp]
aux = [os.path.join(W,'tex',n) for n in os.listdir(W+'/tex') if n.endswith(('.png','.hdr','.obj'))] + [TMP+'/cube.obj']
shutil.copy(ROOT+'/scenes/model/cube.obj',TMP+'/cube.obj')
nbad=0; ncase=0
def mutate(data):
    data=bytearray(data); k=random.random()
    if k<0.4 and len(data)>4: data=data[:random.randrange(1,len(data))]
    elif k<0.8:
        for _ in range(random.randrange(1,8)):
            if data: data[random.randrange(len(data))]=random.randrange(256)
    else:
        i=random.randrange(len(data)+1); data[i:i]=bytes(random.randrange(256) for _ in range(random.randrange(1,40)))
    return bytes(data)
for f in targets:
    src=open(f,'rb').read(); d=os.path.dirname(os.path.abspath(f))
    for i in range(N):
        m=os.path.join(d,'_fuzz_tmp.txt'); open(m,'wb').write(mutate(src))
        try: bad,r=run(m, cwd=ROOT)
        except subprocess.TimeoutExpired: bad=True; r=None; print('TIMEOUT', f, i)
        ncase+=1
        if bad:
            nbad+=1; print('BAD scene', f, i, (r.stderr[:700] if r else '')); shutil.copy(m, TMP+'/bad_%d.txt'%nbad)
        os.remove(m)
# OBJ fuzz through a tiny scene
open(TMP+'/objscene.txt','w').write("MATERIAL m\nTYPE Lambertian\nALBEDO .5 .5 .5\n\nCAMERA\nRES 8 8\nFOVY 20\nITERATIONS 1\nDEPTH 2\nFILE x\nEYE 0 0 5\nLOOKAT 0 0 0\nUP 0 1 0\n\nOBJECT o\n"+TMP+"/cube_f.obj\nmaterial m\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n")
orig=open(TMP+'/cube.obj','rb').read()
for i in range(N*2):
    open(TMP+'/cube_f.obj','wb').write(mutate(orig))
    try: bad,r=run(TMP+'/objscene.txt')
    except subprocess.TimeoutExpired: bad=True; r=None; print('TIMEOUT obj', i)
    ncase+=1
    if bad: nbad+=1; print('BAD obj', i, (r.stderr[:700] if r else '')); shutil.copy(TMP+'/cube_f.obj',TMP+'/bad_%d.obj'%nbad)
for a in aux[:-1]:
    orig=open(a,'rb').read()
    for i in range(N):
        open(a,'wb').write(mutate(orig))
        try: bad,r=run(sp)
        except subprocess.TimeoutExpired: bad=True; r=None; print('TIMEOUT aux', a, i)
        ncase+=1
        if bad: nbad+=1; print('BAD aux', a, i, (r.stderr[:800] if r else '')); shutil.copy(a, TMP+'/bad_%d_%s'%(nbad, os.path.basename(a)))
    open(a,'wb').write(orig)
print('cases', ncase, 'bad', nbad, ('(inputs kept in ' + TMP + ')') if nbad else '')
if not nbad: shutil.rmtree(TMP, ignore_errors=True)
sys.exit(1 if nbad else 0)
