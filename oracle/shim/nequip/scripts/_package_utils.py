def register_libraries_as_external_for_packaging(extern_modules=None, **kwargs):
    return None
