from .._overlay import extend_path

extend_path(__path__, "mvs")
