import re
class Cigar:
    def __init__(self, s): self.s = s
    def items(self):
        for n, op in re.findall(r'(\d+)([MIDNSHP=X])', self.s):
            yield int(n), op
    def __len__(self):
        return sum(n for n, op in self.items() if op in 'MIS=X')
