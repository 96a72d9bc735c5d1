class heapdict(dict): pass
