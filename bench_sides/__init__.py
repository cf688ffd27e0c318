"""side measurements of bench.py, one module per family; bench.py holds the headline step, the rank plumbing and the output line"""
