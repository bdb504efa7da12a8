$1=="GRADREL" { if ($3 > 3e-4 || $3 > 20*$4) print; next } { print }
