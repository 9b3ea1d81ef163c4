module sfgwas_lattigo_fixtures

go 1.18

// the exact replace line of the reference (go.mod:5) and its lattigo requirement (go.mod:12)
replace github.com/ldsec/lattigo/v2 => github.com/hcholab/lattigo/v2 v2.1.2-0.20230123224332-e8d68c24b94a

require github.com/ldsec/lattigo/v2 v2.4.0
