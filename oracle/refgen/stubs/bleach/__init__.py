def clean(s, *a, **k):
    return s.replace("<", "&lt;").replace(">", "&gt;")
