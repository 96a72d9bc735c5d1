_T = str.maketrans('ACGTUNacgtunRYKMBVDHrykmbvdhSWsw', 'TGCAANtgcaanYRMKVBHDyrmkvbhdSWsw')
class Seq:
    def __init__(self, s): self._s = str(s)
    def reverse_complement(self): return Seq(self._s.translate(_T)[::-1])
    def complement(self): return Seq(self._s.translate(_T))
    def __str__(self): return self._s
    def __len__(self): return len(self._s)
    def __getitem__(self, k): return Seq(self._s[k])
