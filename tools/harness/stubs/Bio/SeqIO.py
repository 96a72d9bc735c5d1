def parse(*a, **k): raise NotImplementedError
