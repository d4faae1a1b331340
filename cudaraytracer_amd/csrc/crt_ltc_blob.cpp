// Embeds data/ltc_tables.f32 (ltc_1 then ltc_2, 64x64x4 float32 each) into the
// library's read-only data.  CRT_LTC_PATH is set by the Makefile.
#ifndef CRT_LTC_PATH
#error "CRT_LTC_PATH must name the ltc_tables.f32 file"
#endif
__asm__(".section .rodata\n"
        ".balign 64\n"
        ".global crt_ltc_tables_blob\n"
        "crt_ltc_tables_blob:\n"
        ".incbin \"" CRT_LTC_PATH "\"\n"
        ".global crt_ltc_tables_blob_end\n"
        "crt_ltc_tables_blob_end:\n"
        ".byte 0\n"
        ".previous\n");
