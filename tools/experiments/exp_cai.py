"""Can torch alias foreign device memory through __cuda_array_interface__ on
this ROCm build, and does it release the owner when the tensor dies?"""
import ctypes, gc, weakref, torch
hip = ctypes.CDLL('libamdhip64.so')
p = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)) == 0


class Block:
    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False),
                                         'version': 2, 'strides': None}

    def __del__(self):
        print('block released')


b = Block(p.value, 1 << 18)
w = weakref.ref(b)
t = torch.as_tensor(b, device='cuda')
print('tensor', t.shape, t.dtype, t.device, hex(t.data_ptr()), hex(p.value))
del b
gc.collect()
print('owner alive while tensor lives:', w() is not None)
t.fill_(3.0)
torch.cuda.synchronize()
v = t.view(512, 512)[3:5]
del t
gc.collect()
print('owner alive while a view lives:', w() is not None)
print(float(v.sum()))
del v
gc.collect()
print('owner alive after all tensors died:', w() is not None)
