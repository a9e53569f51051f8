"""Does k_ovo_fused's duration at C2 follow HOW the memory under the output planes was allocated?  Planes from torch.empty (hipMalloc
through torch's caching allocator), from hipMalloc directly, from hipExtMallocWithFlags(hipDeviceMallocContiguous) and from the
virtual-memory API (one physical allocation mapped at once).  The kernel is timed with the engine's own HIP events."""
import ctypes, sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd import _lib
from illico_amd._lib import Engine
N, M, G = 300000, 8000, 2000
dev = torch.device("cuda:0")
X = bench.make_matrix(torch, N, M, 0.5, 0, dev)
eng = Engine(0); eng.set_groups(bench.group_container(bench.make_labels(N, G, 0), G, False))
hip = ctypes.CDLL("libamdhip64.so")
plane_bytes = G * M * 8

def run(ptrs, tag, reps=12):
    flags = _lib.FLAG_INPUT_DEVICE | _lib.FLAG_OUTPUT_DEVICE | _lib.FLAG_DEFER | eng._flags(False, True, True)
    def call():
        eng._check(eng.lib.illico_run_dense(eng.h, X.data_ptr(), _lib.dtype_code("float32"), N, M, X.stride(0), 0, M, flags, 0, ptrs[0], ptrs[1], ptrs[2], M))
    for _ in range(3): call()
    eng.synchronize(); eng.profile(True); eng.profile_reset()
    for _ in range(reps): call()
    eng.synchronize(); p = eng.profile_get(); eng.profile(False)
    k = p["k_ovo_fused"]
    print(f"{tag:60s} k_ovo_fused {k['ms'] / k['launches']:.4f} ms", flush=True)

for rep in range(2):
    t = [torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3)]
    run([x.data_ptr() for x in t], f"torch.empty #{rep}")
    globals()[f"keep{rep}"] = t
for rep in range(2):
    ps = []
    for _ in range(3):
        p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(plane_bytes)) == 0; ps.append(p.value)
    run(ps, f"hipMalloc, three allocations #{rep}")
p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(3 * plane_bytes)) == 0
run([p.value + k * plane_bytes for k in range(3)], "hipMalloc, one allocation")
for rep in range(2):
    t3 = torch.empty((3, G, M), dtype=torch.float64, device=dev)
    run([t3[k].data_ptr() for k in range(3)], f"torch.empty((3, G, M)) #{rep}")
    globals()[f"keep3_{rep}"] = t3
for pad in (4096, 65536 + 4096, (1 << 21) + 4096):
    ps = []
    for k in range(3):
        p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(plane_bytes + 3 * pad)) == 0; ps.append(p.value + k * pad)
    run(ps, f"hipMalloc, three allocations, plane k shifted by k * {pad} B")
    print("   bases mod 2^21:", [hex(x % (1 << 21)) for x in ps], " >>21:", [x >> 21 for x in ps])
for rep in range(2):
    p = ctypes.c_void_p(); rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(3 * plane_bytes), ctypes.c_uint(0x4))
    if rc != 0: print("hipDeviceMallocContiguous failed", rc); break
    run([p.value + k * plane_bytes for k in range(3)], f"hipExtMallocWithFlags(Contiguous), one allocation #{rep}")
# virtual-memory API: one physical handle of 3 planes (rounded to the granularity), mapped into one reserved range
class Prop(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("loc_type", ctypes.c_int), ("loc_id", ctypes.c_int),
                ("win32", ctypes.c_void_p), ("allocFlags", ctypes.c_ubyte * 8)]
prop = Prop(); prop.type = 1  # hipMemAllocationTypePinned
prop.loc_type = 1; prop.loc_id = 0  # hipMemLocationTypeDevice
gran = ctypes.c_size_t()
rc = hip.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), ctypes.c_int(1))  # recommended
print("granularity rc", rc, gran.value)
if rc == 0 and gran.value:
    size = (3 * plane_bytes + gran.value - 1) // gran.value * gran.value
    h = ctypes.c_void_p(); rc = hip.hipMemCreate(ctypes.byref(h), ctypes.c_size_t(size), ctypes.byref(prop), ctypes.c_ulonglong(0)); print("create", rc)
    va = ctypes.c_void_p(); rc2 = hip.hipMemAddressReserve(ctypes.byref(va), ctypes.c_size_t(size), ctypes.c_size_t(0), ctypes.c_void_p(0), ctypes.c_ulonglong(0)); print("reserve", rc2)
    rc3 = hip.hipMemMap(va, ctypes.c_size_t(size), ctypes.c_size_t(0), h, ctypes.c_ulonglong(0)); print("map", rc3)
    class Acc(ctypes.Structure):
        _fields_ = [("loc_type", ctypes.c_int), ("loc_id", ctypes.c_int), ("flags", ctypes.c_int)]
    acc = Acc(1, 0, 3)
    rc4 = hip.hipMemSetAccess(va, ctypes.c_size_t(size), ctypes.byref(acc), ctypes.c_size_t(1)); print("access", rc4)
    if rc == 0 and rc2 == 0 and rc3 == 0 and rc4 == 0:
        run([va.value + k * plane_bytes for k in range(3)], "hipMemCreate + hipMemMap, one physical allocation")
