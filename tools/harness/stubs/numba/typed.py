class Dict(dict):
    @classmethod
    def empty(cls, key_type=None, value_type=None):
        return cls()
class List(list):
    @classmethod
    def empty_list(cls, item_type=None):
        return cls()
    def __getitem__(self, k):
        r = list.__getitem__(self, k)
        if isinstance(k, slice):
            return List(r)
        return r
    def copy(self):
        return List(self)
